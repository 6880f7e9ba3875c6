#!/bin/bash
# PMC passes (own runs, --kernel-trace only) of the QPLEX learner: MFMA busy / HBM bytes of the fused head kernels.
#   gpurun -- 'bash tools/pmc_qplex.sh r01'   then   python tools/summarize_profiles.py r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/${TAG}_pmcq
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmcq/pass$i -o p -- python3 tools/prof_learner.py --alg qplex --shape 2s3z --envs 4096 --updates 3 > $OUT/${TAG}_pmcq.pass$i.log 2>&1 || true
done
ls $OUT/${TAG}_pmcq/*
