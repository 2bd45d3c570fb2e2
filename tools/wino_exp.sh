#!/bin/bash
# Development aid: experiment / timing builds of K17 next to the product library (FAR_WINO_EXP bit mask: 1 no transform, 2 no MFMAs,
# 4 no weight requests, 8 no raw requests, 16 no epilogue, 32 draining waits), built HERE (cross-compile) into far_amd/lib/exp/ so that
# they travel to the GPU box; run there with tools/wino_exp_run.sh.
cd "$(dirname "$0")/.."
mkdir -p far_amd/lib/exp
OBJS=$(ls far_amd/lib/*.o | grep -v conv_wino_f16s.o)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -I far_amd/csrc"
for e in ${FAR_WINO_EXPS:-1 2 3 4 8 12 16 32}; do
  ( /opt/rocm/bin/hipcc $FL -DFAR_WINO_EXP=$e -c far_amd/csrc/conv_wino_f16s.hip -o far_amd/lib/exp/wino_exp$e.o.tmp && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o far_amd/lib/exp/libfar_exp$e.so $OBJS far_amd/lib/exp/wino_exp$e.o.tmp ) &
done
( /opt/rocm/bin/hipcc $FL -DFAR_WINO_TIMING -c far_amd/csrc/conv_wino_f16s.hip -o far_amd/lib/exp/wino_timing.o.tmp && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o far_amd/lib/exp/libfar_timing.so $OBJS far_amd/lib/exp/wino_timing.o.tmp ) &
for e in ${FAR_WINO_TEXPS:-}; do
  ( /opt/rocm/bin/hipcc $FL -DFAR_WINO_TIMING -DFAR_WINO_EXP=$e -c far_amd/csrc/conv_wino_f16s.hip -o far_amd/lib/exp/wino_texp$e.o.tmp && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o far_amd/lib/exp/libfar_timing_exp$e.so $OBJS far_amd/lib/exp/wino_texp$e.o.tmp ) &
done
wait
rm -f far_amd/lib/exp/*.tmp
ls -la far_amd/lib/exp
