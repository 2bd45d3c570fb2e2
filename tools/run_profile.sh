#!/bin/bash
# Profiling recipe for profiles/rNN_bench_fp32_kernel_trace.txt (run on the GPU box through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/run_profile.sh'
# then here: python tools/make_profile_txt.py
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-modes > gpurun_out/bench_prof.log 2>&1
python3 tools/profile_report.py $(ls /tmp/prof/*.db /tmp/prof/*/*.db 2>/dev/null | head -1) > gpurun_out/r01c_trace.txt 2>&1
# HBM traffic (separate counter passes, MI355X_MICROARCH.md "HBM"): kernels in isolation at the bench shapes
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- python3 tools/kprobe.py all 32 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- python3 tools/kprobe.py all 32 2 > /dev/null 2>&1
python3 tools/pmc_traffic.py $(ls /tmp/pf/*.db /tmp/pf/*/*.db 2>/dev/null | head -1) $(ls /tmp/pw/*.db /tmp/pw/*/*.db 2>/dev/null | head -1) > gpurun_out/r01_pmc_traffic.json 2> gpurun_out/pmc_err.log
tail -1 gpurun_out/bench_plain.log | cut -c1-200
