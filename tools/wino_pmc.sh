#!/bin/bash
# Runs on the GPU box: counter passes (separate runs, as MI355X_MICROARCH.md prescribes) over tools/wino_probe.py; prints per-kernel
# averages with tools/pmc_util.py.  Usage: bash tools/wino_pmc.sh [probe args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd $R; timeout 200 rocprofv3 --pmc $set -d /tmp/wpmc$i -o p -- python3 tools/wino_probe.py "$@" > /tmp/wpmc$i.log 2>&1)
  echo "== $set"
  (cd $R; python tools/pmc_util.py /tmp/wpmc$i/p_results.db k_wino k_conv 2>&1 | grep -v "^#" | cut -c1-400)
done
