#!/bin/bash
# rocprofv3 counter passes for the learner (run on the GPU box from the repo root):
#   tools/pmc_passes.sh <outdir> [prof_learner.py args...]
# Counters are collected in their own runs (no trace domains besides --kernel-trace), one pass per
# counter group, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 tools/prof_learner.py "$@" > "$OUT.pass$i.log" 2>&1 || true
done
ls "$OUT"/*/*/ | head -30
