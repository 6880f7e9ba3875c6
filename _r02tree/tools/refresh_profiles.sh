#!/bin/bash
# Regenerates the evidence under profiles/ (run ON THE GPU BOX from the repo root through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# 1. rocprofv3 --kernel-trace --stats of the default bench command           -> gpurun_out/<tag>_bench/
# 2. PMC passes (own runs, --kernel-trace only) of the learner               -> gpurun_out/<tag>_pmc/pass{1..4}
# 3. in-kernel stamp shares of the three persistent kernels (diagnostic .so) -> gpurun_out/<tag>_stamps.txt
# Then, in the container: python tools/summarize_profiles.py <tag>   (writes profiles/<tag>_*)
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/${TAG}_bench $OUT/${TAG}_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bench -o p -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1
grep '^{"metric"' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench_line.json
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmc/pass$i -o p -- python3 tools/prof_learner.py --updates 3 --rollouts 2 > $OUT/${TAG}_pmc.pass$i.log 2>&1 || true
done
make -C marl_amd/csrc stamps > /dev/null 2>&1
( for k in rollout fwd bwd wgrad qmix; do python3 tools/stamps.py $k 4096 2>/dev/null | grep -v amdgpu.ids; echo; done
  for k in fwd_pipe bwd_pipe rollout; do python3 tools/stamps.py $k 512 2>/dev/null | grep -v amdgpu.ids; echo; done ) > $OUT/${TAG}_stamps.txt
python3 bench.py > $OUT/${TAG}_bench_full.log 2>&1
grep '^{"metric"' $OUT/${TAG}_bench_full.log | tail -1 > $OUT/${TAG}_bench_full_line.json
# 4. QPLEX learner (config 3 shape, 4096 envs): kernel stats + standalone timings of the fused head kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qplex -o p -- python3 tools/prof_learner.py --alg qplex --shape 2s3z --envs 4096 --updates 6 > $OUT/${TAG}_qplex.log 2>&1
python3 tools/time_mlp3.py 4096 > $OUT/${TAG}_mlp3_times.txt 2>&1
ls $OUT/${TAG}_bench $OUT/${TAG}_pmc/* | head -40
