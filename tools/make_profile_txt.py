"""Assembles profiles/r01_bench_fp32_kernel_trace.txt from the files a profiling gpurun call leaves in gpurun_out/:
  bench_prof.log (bench.py under rocprofv3), r01c_trace.txt (tools/profile_report.py on the rocpd database),
  bench_plain.log (the un-profiled python bench.py)."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
last_json = lambda path: [l for l in open(path).read().splitlines() if l.startswith('{')][-1]
plain=last_json(ROOT + '/gpurun_out/bench_plain.log')
prof=last_json(ROOT + '/gpurun_out/bench_prof.log')
tr=open(ROOT + '/gpurun_out/r01c_trace.txt').read()
whole,win=tr.split('\n\n',1)
out=f"""# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline    (round 1, 1x MI355X, default fp32-grade path)
# bench line of the same profiled run:
{prof}

# un-profiled bench line (python bench.py, i.e. 5 steps after 3 warm-up steps) on the same box:
{plain}

## one steady-state step (32 pairs): per kernel
{win.strip()}

## whole process (includes the warm-up steps and the isolated kernel timings of bench.kernel_rooflines)
{whole.strip()}
"""
open(ROOT + '/profiles/r01_bench_fp32_kernel_trace.txt','w').write(out)
