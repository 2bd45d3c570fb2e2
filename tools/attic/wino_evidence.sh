#!/bin/bash
# Runs on the GPU box (after tools/wino_exp.sh built the experiment libraries here): everything the K17 go / no-go statement of
# DESIGN section 7 rests on, into gpurun_out/r04_k17_evidence.txt -- errors vs float64 next to K9's, same-box timings at the bench
# shapes, the experiment decomposition (FAR_WINO_EXP masks), the in-kernel timeline, and the counter passes.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/r04_k17_evidence.txt
{
echo "# K17 (Winograd F(2x2,3x3), split fp16) evidence; commit ${K17_COMMIT:-unknown}"
echo "## errors (max / rms relative to max|ref| resp. rms(ref) of a float64 convolution) and timings"
timeout 600 python tools/wino_ab.py 2>&1 | grep -v amdgpu.ids
echo
echo "## experiment builds (timings only; mask bits: 1 no input transform, 2 no MFMAs, 4 no weight requests, 8 no raw-patch requests, 16 no epilogue, 32 draining waits)"
for f in far_amd/lib/exp/libfar_exp*.so; do
  echo "== $(basename $f)"
  WINO_NOCHECK=1 FAR_HIP_LIB=$PWD/$f timeout 300 python tools/wino_ab.py --quick 2>&1 | tail -1
done
echo
echo "## in-kernel timeline (s_memtime stamps, cycles at 100 MHz x 18.2; library built with -DFAR_WINO_TIMING)"
FAR_HIP_LIB=$PWD/far_amd/lib/exp/libfar_timing.so timeout 300 python tools/wino_timing.py 2>&1 | grep -v amdgpu.ids
echo
echo "## counters (separate passes)"
bash tools/wino_pmc.sh 2>&1
} > $O 2>&1
tail -5 $O
