#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for f in far_amd/lib/exp/libfar_exp*.so; do
  echo "== $f"
  WINO_NOCHECK=1 FAR_HIP_LIB=$PWD/$f python tools/wino_ab.py --quick 2>&1 | tail -1
done
