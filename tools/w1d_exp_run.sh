#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
W1D_NOCHECK=1 python tools/wino1d_ab.py --quick 2>&1 | tail -1
for f in far_amd/lib/exp/libfar_w1dexp*.so; do
  echo "== $f"
  W1D_NOCHECK=1 FAR_HIP_LIB=$PWD/$f python tools/wino1d_ab.py --quick 2>&1 | tail -1
done
