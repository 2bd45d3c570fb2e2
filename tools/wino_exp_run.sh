#!/bin/bash
# Runs on the GPU box: product timing, the s_memtime timeline, then each experiment library (timings only: their results are wrong).
cd "${GRAFT_REPO_ROOT:-.}"
python tools/wino_ab.py --quick 2>&1 | tail -2
FAR_HIP_LIB=$PWD/far_amd/lib/exp/libfar_timing.so python tools/wino_timing.py 2>&1 | grep -v amdgpu.ids
for f in far_amd/lib/exp/libfar_exp*.so; do
  echo "== $f"
  FAR_HIP_LIB=$PWD/$f python tools/wino_ab.py --quick 2>&1 | tail -1
done
