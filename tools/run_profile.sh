#!/bin/bash
# Profiling recipe for profiles/rNN_bench_fp32_kernel_trace.txt (run on the GPU box through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/run_profile.sh'
# then here: python tools/make_profile_txt.py
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
python3 tools/profile_report.py $(ls /tmp/prof/*.db /tmp/prof/*/*.db 2>/dev/null | head -1) > gpurun_out/r01c_trace.txt 2>&1
tail -1 gpurun_out/bench_plain.log | cut -c1-200
