"""Host-side view of the training step: torch.profiler (CPU activities) over a few steps of bench.py --workload c3, top operators
by self CPU time -- autograd nodes of far_amd.ops show up under their Function names.  python tools/c3_host_profile.py  (GPU box)"""
import os
import runpy
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py', '--workload', 'c3', '--no-cpu-baseline', '--no-other-modes', '--steps', '6', '--warmup', '4']
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=60))
