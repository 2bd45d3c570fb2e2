#!/bin/bash
# Runs on the GPU box: the step's per-family times (tools/step_floors.py) with the shipped library, then with an experiment build
# -DFAR_STAGGER_F16S (static issue-priority stagger by wave slot in K1 / K2 / K9 / K13 / K14), same box.  Usage: bash tools/stagger_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/step_floors.py --out gpurun_out/sf_base.txt > /dev/null 2>&1
export FAR_EXTRA_HIPCC_FLAGS="-DFAR_STAGGER_F16S"
python -m far_amd.build > gpurun_out/stagger_build.log 2>&1
python tools/step_floors.py --out gpurun_out/sf_stagger.txt > /dev/null 2>&1
for f in gpurun_out/sf_base.txt gpurun_out/sf_stagger.txt; do echo "== $f"; grep "step time" $f; sed -n '/by kernel family/,/launches ranked/p' $f | head -12; done
