#!/bin/bash
# Development aid: builds experiment variants of K9 (FAR_K9_EXP bit mask: 1 = no activation loads, 2 = no epilogue) next to
# the product library and times the step's K9 shapes with each (tools/k9_ab.py).  Run on the GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
OBJS=$(ls far_amd/lib/*.o | grep -v conv_igemm_f16s.o)
for e in ${FAR_K9_EXPS:-1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -I far_amd/csrc -DFAR_K9_EXP=$e -c far_amd/csrc/conv_igemm_f16s.hip -o /tmp/conv_exp$e.o &
done
wait
python tools/k9_ab.py 2>/dev/null | head -12
for e in ${FAR_K9_EXPS:-1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libfar_exp$e.so $OBJS /tmp/conv_exp$e.o
  python tools/k9_ab.py /tmp/libfar_exp$e.so 2>/dev/null | head -12
done
