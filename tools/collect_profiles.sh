#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of bench.py, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes as
# MI355X_MICROARCH.md prescribes) on the isolated kernels, the un-profiled bench lines.  Output under gpurun_out/; the
# summaries that are judged get copied into profiles/ by tools/make_profile_txt.py (run in the build container).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
mkdir -p $R/gpurun_out
# page the image in first (the first process on a fresh box runs with a slow host for a while: 107-113 ms steps were measured there
# against 94-95 ms in every later process), then the un-profiled headline line: counter collection can leave the clocks in the
# profiler's fixed state for a while
(cd $R; timeout 300 python tools/step_times.py 3 > /dev/null 2>&1)
(cd $R; timeout 400 python bench.py > gpurun_out/bench_plain.log 2> gpurun_out/bench_plain.err)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-modes --no-other-workloads > $R/gpurun_out/bench_prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp16 -o bench -- python3 $R/bench.py --precision fp16 --steps 3 --warmup 3 --no-cpu-baseline --no-other-modes --no-other-workloads --skip-rooflines > $R/gpurun_out/bench_fp16_prof.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/tools/kprobe.py pmc 32 2 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w -- python3 $R/tools/kprobe.py pmc 32 2 > $R/gpurun_out/pmc_write.log 2>&1
cd $R
timeout 300 python bench.py --workload c4 > gpurun_out/bench_c4.log 2> gpurun_out/bench_c4.err
timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-other-modes --no-other-workloads > gpurun_out/bench_c3.log 2> gpurun_out/bench_c3.err
timeout 300 python bench.py --workload c3 --vendor-train --no-cpu-baseline --no-other-modes --no-other-workloads > gpurun_out/bench_c3_vendor.log 2> gpurun_out/bench_c3_vendor.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-other-modes --no-other-workloads > gpurun_out/bench_c5.log 2> gpurun_out/bench_c5.err
timeout 200 python tools/wgrad_time.py > gpurun_out/${TAG}_wgrad_time.txt 2>&1
timeout 300 python tools/stride_ab.py > gpurun_out/${TAG}_stride_ab.json 2> gpurun_out/stride_ab.err
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python3 $R/bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-other-workloads > $R/gpurun_out/bench_c3_prof.log 2>&1)
python tools/profile_report.py /tmp/prof_c3/c3_results.db > gpurun_out/${TAG}_c3_trace.txt 2>&1
python tools/aten_in_step.py /tmp/prof_c3/c3_results.db > gpurun_out/${TAG}_c3_vendor_kernels.txt 2>&1
python tools/profile_report.py gpurun_out/prof_bench/bench_results.db > gpurun_out/${TAG}_trace.txt 2>&1
python tools/profile_report.py gpurun_out/prof_fp16/bench_results.db > gpurun_out/${TAG}_trace_fp16.txt 2>&1
FAR_COMMIT=${FAR_COMMIT:-unknown} python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db > gpurun_out/${TAG}_pmc_traffic.json 2> gpurun_out/pmc_err.log
rm -rf gpurun_out/prof_bench gpurun_out/prof_fp16 gpurun_out/pmc_fetch gpurun_out/pmc_write      # the databases exceed gpurun's 64 MiB copy-back limit
tail -c 300 gpurun_out/bench_plain.log; echo; head -c 300 gpurun_out/${TAG}_pmc_traffic.json
