#!/bin/bash
# Development aid: experiment builds of K18 next to the product library (FAR_W1D_EXP bit mask: 1 no transform, 2 no MFMAs, 4 no weight
# requests, 8 no raw requests, 16 no epilogue, 32 draining waits), built HERE into far_amd/lib/exp/ so that they travel to the GPU box.
cd "$(dirname "$0")/.."
mkdir -p far_amd/lib/exp
OBJS=$(ls far_amd/lib/*.o | grep -v conv_wino1d_f16s.o)
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -I far_amd/csrc"
for e in ${FAR_W1D_EXPS:-1 2 3 16 18 14 15}; do
  ( /opt/rocm/bin/hipcc $FL -DFAR_W1D_EXP=$e ${FAR_W1D_DEFS:-} -c far_amd/csrc/conv_wino1d_f16s.hip -o far_amd/lib/exp/w1d_exp$e.o.tmp && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o far_amd/lib/exp/libfar_w1dexp$e.so $OBJS far_amd/lib/exp/w1d_exp$e.o.tmp ) &
done
wait
rm -f far_amd/lib/exp/*.tmp
ls far_amd/lib/exp
