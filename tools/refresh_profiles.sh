#!/bin/bash
# Evidence of one round, ON THE GPU BOX from the repo root:   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r06'
# then in the container:   python tools/summarize_profiles.py r06     (writes profiles/r06_*)
#  1. rocprofv3 --kernel-trace --stats of the default bench command (headline only)           -> gpurun_out/<round>_bench/
#  2. PMC passes (own runs, --kernel-trace only; FETCH_SIZE and WRITE_SIZE in separate passes) of the learner at the headline
#     workload AND at every configs[] leg of bench.py - gpurun_out/<round>_pmc_<workload>/pass*/; SQ group (MFMA busy, VALU) for the
#     4096-env workloads
#  3. the full default bench line, soak runs (2000 steps at 4096 and 512 envs), shard steps, learner rates, the fused-head timings
# PARTS (second argument, default "all"): any of  bench pmc stats times rates shards soak line
TAG=${1:?usage: refresh_profiles.sh <round tag, e.g. r06> [parts]}
PARTS=${2:-all}
want() { [ "$PARTS" = all ] || [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
python3 -c "from marl_amd import _lib; print(_lib.load().marl_hip_version().decode())" > $OUT/${TAG}_lib_version.txt
if want bench; then rm -rf $OUT/${TAG}_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bench -o p -- python3 bench.py --no-cpu-baseline --no-configs > $OUT/${TAG}_bench.log 2>&1
grep '^{"metric"' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench_profiled_line.json
fi
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
pmc() {   # workload-name  full(0/1)  prof_learner args...
  local W=$1 FULL=$2; shift 2
  local i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "$P1"; do
    i=$((i+1))
    if [ $i = 3 ] && [ $FULL = 0 ]; then break; fi
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$W/pass$i -o p -- python3 tools/prof_learner.py --updates 3 --warmup 1 "$@" > $OUT/${TAG}_pmc_$W.pass$i.log 2>&1 || true
  done
}
if want pmc; then rm -rf $OUT/${TAG}_pmc_*
pmc qmix_2s3z_T120_envs4096 1 --rollouts 2
pmc qmix_2s3z_T120_envs1024 0 --envs 1024
pmc qplex_2s3z_T120_envs512 0 --alg qplex --envs 512
pmc qtran_base_3s5z_T150_envs512 0 --alg qtran_base --shape 3s5z --envs 512
pmc qmix_MMM2_T120_envs1024_bf16mixer 0 --shape MMM2 --envs 1024 --mixer-dtype bf16
pmc qplex_2s3z_T120_envs512_bf16x6 0 --alg qplex --envs 512 --gemm-mode bf16x6
pmc qplex_2s3z_T120_envs4096 0 --alg qplex --envs 4096
pmc qplex_2s3z_T120_envs4096_bf16x6 1 --alg qplex --envs 4096 --gemm-mode bf16x6
pmc qmix_2s3z_T120_envs4096_bf16x6 1 --gemm-mode bf16x6
pmc qmix_2s3z_T120_envs1024_bf16x6 0 --envs 1024 --gemm-mode bf16x6
pmc qmix_2s3z_T120_envs512 0 --envs 512
pmc qmix_2s3z_T120_envs512_bf16x6 0 --envs 512 --gemm-mode bf16x6
pmc qtran_base_3s5z_T150_envs512_bf16x6 0 --alg qtran_base --shape 3s5z --envs 512 --gemm-mode bf16x6
pmc qmix_MMM2_T120_envs1024_bf16mixer_bf16x6 0 --shape MMM2 --envs 1024 --mixer-dtype bf16 --gemm-mode bf16x6
fi
if want stats; then
# kernel stats of the QMIX / QPLEX updates in both modes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qmix_bf16x6 -o p -- python3 tools/prof_learner.py --alg qmix --envs 4096 --updates 6 --warmup 2 --gemm-mode bf16x6 > $OUT/${TAG}_qmix_bf16x6.log 2>&1
for M in f32 bf16x6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qplex_$M -o p -- python3 tools/prof_learner.py --alg qplex --envs 4096 --updates 6 --warmup 2 --gemm-mode $M > $OUT/${TAG}_qplex_$M.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qtran -o p -- python3 tools/prof_learner.py --alg qtran_base --shape 3s5z --envs 512 --updates 12 --warmup 3 > $OUT/${TAG}_qtran.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_mmm2_bf16 -o p -- python3 tools/prof_learner.py --shape MMM2 --envs 1024 --mixer-dtype bf16 --updates 8 --warmup 3 > $OUT/${TAG}_mmm2_bf16.log 2>&1
fi
if want times; then
# timings
python3 tools/time_mlp3.py 4096 > $OUT/${TAG}_mlp3_times.txt 2>&1
python3 tools/time_unroll_x6.py 4096 1024 512 256 > $OUT/${TAG}_unroll_x6_times.txt 2>&1
SHAPE=3s5z python3 tools/time_unroll_x6.py 512 2048 >> $OUT/${TAG}_unroll_x6_times.txt 2>&1
SHAPE=MMM2 python3 tools/time_unroll_x6.py 1024 >> $OUT/${TAG}_unroll_x6_times.txt 2>&1
python3 tools/time_qmix.py 4096 1024 512 > $OUT/${TAG}_qmix_times.txt 2>&1
python3 tools/time_rollout.py 2>&1 | grep -v "Init\|amdgpu" > $OUT/${TAG}_rollout_times.txt
( echo "== 3s5z"; SHAPE=3s5z python3 tools/time_rollout.py 2048 1536 512; echo "== MMM2"; SHAPE=MMM2 python3 tools/time_rollout.py 1024 ) 2>&1 | grep -v "Init\|amdgpu" >> $OUT/${TAG}_rollout_times.txt
( echo "== whole rollout, round-6 kernel (csrc/rollout_x6.hip)"; python3 tools/stamps_rollout_x6.py 4096; python3 tools/stamps_rollout_x6.py 2048
  echo "== whole rollout, round-5 kernel (csrc/rollout_x6_v1.hip)"; MARL_ROLLOUT_V1=1 python3 tools/stamps_rollout_x6.py 4096; MARL_ROLLOUT_V1=1 python3 tools/stamps_rollout_x6.py 512
  echo "== plain unroll, round-6 decomposition (csrc/agent_x6p.hip)"; python3 tools/stamps_unroll_x6p.py 4096; python3 tools/stamps_unroll_x6p.py 2048 ) 2>&1 | grep -v "Init\|amdgpu" > $OUT/${TAG}_stamps.txt
fi
if want rates; then
( for a in "--alg qmix --envs 512" "--alg qmix --envs 4096" "--alg qplex --envs 512" "--alg qplex --envs 4096"; do for w in 1 100000; do
    echo -n "$a --gemm-mode bf16x6, split BPTT $([ $w = 1 ] && echo on || echo off) : "; MARL_X6_BWD_MIN_WG=$w python3 tools/prof_learner.py $a --gemm-mode bf16x6 --warmup 5 --updates 30 2>/dev/null | tail -1; done; done ) > $OUT/${TAG}_bptt_x6_ab.txt
( for a in "--alg qmix --envs 1024" "--alg qmix --envs 4096" "--alg vdn --envs 4096" "--alg qplex --envs 512" "--alg qplex --envs 512 --gemm-mode bf16x6" "--alg qplex --envs 4096" \
           "--alg qplex --envs 4096 --gemm-mode bf16x6" "--alg qmix --envs 512" "--alg qmix --envs 512 --gemm-mode bf16x6" "--alg qmix --envs 1024 --gemm-mode bf16x6" "--alg qmix --envs 4096 --gemm-mode bf16x6" "--alg qtran_base --shape 3s5z --envs 512" "--alg qtran_base --shape 3s5z --envs 512 --gemm-mode bf16x6" "--alg qtran_base --shape 3s5z --envs 2048" "--shape MMM2 --envs 1024" "--shape MMM2 --envs 1024 --mixer-dtype bf16" "--shape MMM2 --envs 1024 --mixer-dtype bf16 --gemm-mode bf16x6" "--alg qmix --shape 3s5z --envs 1024" "--alg qmix --shape 3s5z --envs 1024 --gemm-mode bf16x6"; do
    echo -n "$a : "; python3 tools/prof_learner.py $a --warmup 5 --updates 20 2>/dev/null | tail -1; done ) > $OUT/${TAG}_learner_rates.txt
fi
if want shards; then
for e in 512 1024 2048 4096; do python3 bench.py --envs $e --gemm-mode f32 --no-cpu-baseline --no-configs --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('envs_per_gpu=%d gemm_mode=f32 hip_graph=%s : ms_per_step %.3f env-steps/s %.2f M  learner updates/s %.1f  rollout M env-steps/s %.1f' % (d['config']['envs_per_gpu'], d['config']['hip_graph'], d['ms_per_step'], d['value']/1e6, d['learner_updates_per_sec'], d['rollout_env_steps_per_sec']/1e6))"; done > $OUT/${TAG}_shard_steps.txt
for e in 512 1024 2048 4096; do python3 bench.py --envs $e --gemm-mode bf16x6 --no-twin --no-cpu-baseline --no-configs --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('envs_per_gpu=%d gemm_mode=bf16x6 hip_graph=%s : ms_per_step %.3f env-steps/s %.2f M  learner updates/s %.1f  rollout M env-steps/s %.1f' % (d['config']['envs_per_gpu'], d['config']['hip_graph'], d['ms_per_step'], d['value']/1e6, d['learner_updates_per_sec'], d['rollout_env_steps_per_sec']/1e6))"; done >> $OUT/${TAG}_shard_steps.txt
fi
if want shards; then   # what a world-1 RCCL reducer (two collectives + the agreed-length read-back per update) adds to a 512-env step
( for r in 0 1; do echo -n "envs_per_gpu=512 force_reducer=$r : "; MARL_FORCE_REDUCER=$r python3 bench.py --envs 512 --no-twin --no-cpu-baseline --no-configs --steps 60 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step %.3f learner updates/s %.1f' % (d['ms_per_step'], d['learner_updates_per_sec']))"; done ) > $OUT/${TAG}_reducer_world1.txt 2>&1
fi
if want soak; then
for e in 4096 512; do python3 bench.py --envs $e --no-twin --no-cpu-baseline --no-configs --steps 2000 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('soak: envs=%d steps=%d ms_per_step %.3f env-steps/s %.2f M (timed %.1f s)' % (d['config']['global_envs'], d['steps'], d['ms_per_step'], d['value']/1e6, d['ms_per_step']*d['steps']/1e3))"; done > $OUT/${TAG}_soak.txt
fi
if want line; then
timeout 1200 python3 bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_full.log
cp bench_full.json $OUT/${TAG}_bench_full.json
fi
ls $OUT | grep ${TAG}_ | head -60
