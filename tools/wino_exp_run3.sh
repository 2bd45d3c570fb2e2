#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for f in far_amd/lib/exp/libfar_timing*.so; do
  echo "== $f"
  FAR_HIP_LIB=$PWD/$f python tools/wino_timing.py 2>&1 | grep -v amdgpu.ids | grep -A7 "^wave 0\|^wave 4" | grep -v "^wave [1235]"
done
