"""Host-side profile of the training step (bench.py --workload c3 under cProfile): where the Python time of a step goes.
python tools/c3_host_profile.py [steps]   (GPU box)"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = sys.argv[1] if len(sys.argv) > 1 else '10'
sys.argv = ['bench.py', '--workload', 'c3', '--no-cpu-baseline', '--no-other-modes', '--steps', steps, '--warmup', '3']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
finally:
    pr.disable()
for key in ('tottime', 'cumulative'):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(28)
    print(out.getvalue()[:6000])
