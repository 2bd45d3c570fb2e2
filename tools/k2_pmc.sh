#!/bin/bash
# Runs on the GPU box: counter passes (separate runs, no trace domains, as MI355X_MICROARCH.md prescribes) over tools/k2_time.py -- what
# K2's P v~ pass (k_pv) and its statistics pass (k_rowstats) do with the matrix pipe.  Usage: bash tools/k2_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  (cd $R; timeout 300 rocprofv3 --pmc $set -d /tmp/k2pmc$i -o p -- python3 tools/k2_time.py > /tmp/k2pmc$i.log 2>&1)
  echo "== $set"
  (cd $R; python tools/pmc_util.py /tmp/k2pmc$i/p_results.db k_pv k_rowstats 2>&1 | grep -v "^#" | cut -c1-500)
done
