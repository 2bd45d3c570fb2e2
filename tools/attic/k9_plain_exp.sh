#!/bin/bash
# Development aid: what bounds K9 with PLAIN fp16 operands (the 'fp16' mode's backbone)?  Builds -DFAR_K9_EXP variants of the kernel
# (1 no activation loads, 2 no epilogue, 16 half the weight-fragment reads, 32 one weight-slab request in eight, 64 no MFMAs; all but
# the first two give wrong results) next to the product library and times the backbone's stride-1 3x3 shapes with each.  GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
OBJS=$(ls far_amd/lib/*.o | grep -v conv_igemm_f16s.o)
EXPS=${FAR_K9_EXPS:-1 2 3 16 32 48 64}
for e in $EXPS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -I far_amd/csrc -DFAR_K9_EXP=$e -c far_amd/csrc/conv_igemm_f16s.hip -o /tmp/conv_exp$e.o &
done
wait
echo "== product"; python tools/k9_plain_time.py 0 2>/dev/null
for e in $EXPS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libfar_exp$e.so $OBJS /tmp/conv_exp$e.o
  echo "== FAR_K9_EXP=$e"; FAR_HIP_LIB=/tmp/libfar_exp$e.so python tools/k9_plain_time.py 0 2>/dev/null
done
