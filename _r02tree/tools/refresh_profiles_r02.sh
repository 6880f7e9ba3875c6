#!/bin/bash
# Round-2 additions to the evidence under profiles/ (run ON THE GPU BOX through gpurun, after refresh_profiles.sh):
#   gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh r02; bash tools/refresh_profiles_r02.sh r02'
# then in the container: python tools/summarize_profiles.py r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
# config 4: QTRAN-base, 3s5z, 512 envs (the per-GPU shard of 2048 envs / 4 GPUs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qtran -o p -- python3 tools/prof_learner.py --alg qtran_base --shape 3s5z --envs 512 --warmup 3 --updates 10 > $OUT/${TAG}_qtran.log 2>&1
# config 5: QMIX, MMM2, 1024 envs (8192 / 8 GPUs), fp32 and bf16 mixer: kernel stats + the bench lines (HBM-bound roofline of the mixer)
for DT in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_mmm2_$DT -o p -- python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 3 --updates 10 --mixer-dtype $DT > $OUT/${TAG}_mmm2_$DT.log 2>&1
  python3 bench.py --shape MMM2 --envs 1024 --mixer-dtype $DT --roofline-kernel mixer --no-cpu-baseline --steps 10 --warmup 3 > $OUT/${TAG}_bench_mmm2_$DT.log 2>&1
  grep '^{"metric"' $OUT/${TAG}_bench_mmm2_$DT.log | tail -1 > $OUT/${TAG}_bench_mmm2_${DT}_line.json
done
# HBM traffic of the wide-state QMIX kernels (separate PMC passes, --kernel-trace only)
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmcw/pass$i -o p -- python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 1 --updates 3 --mixer-dtype bf16 > $OUT/${TAG}_pmcw.pass$i.log 2>&1 || true
done
# learner rates of every BASELINE configuration at its per-GPU size
( for C in "qmix 2s3z 1024" "qmix 2s3z 4096" "vdn 2s3z 4096" "qplex 2s3z 512" "qplex 2s3z 4096" "qtran_base 3s5z 512" "qtran_base 3s5z 2048" "qmix MMM2 1024"; do
    set -- $C
    echo -n "$1 $2 envs=$3 : "; python3 tools/prof_learner.py --alg $1 --shape $2 --envs $3 --warmup 5 --updates 20 2>/dev/null | grep updates
  done
  echo -n "qmix MMM2 envs=1024 mixer bf16 : "; python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 5 --updates 20 --mixer-dtype bf16 2>/dev/null | grep updates ) > $OUT/${TAG}_learner_rates.txt
# single-GPU step times at the shard sizes of the 2 / 4 / 8-GPU strong-scaling runs
( for E in 512 1024 2048 4096; do
    python3 bench.py --envs $E --steps 20 --warmup 6 --no-cpu-baseline $SHARD_FLAGS 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('envs_per_gpu=%d : ms_per_step %.3f env-steps/s %.2f M  learner updates/s %.1f  rollout M env-steps/s %.1f' % (d['config']['global_envs'], d['ms_per_step'], d['value']/1e6, d['learner_updates_per_sec'], d['rollout_env_steps_per_sec']/1e6))"
  done ) > $OUT/${TAG}_shard_steps.txt
# hardware probes: HBM read ceiling; fp32 MFMA vs VALU / LDS issue on one SIMD (built in the container: tools/probe/)
( [ -x tools/probe/bw_probe ] && timeout 120 tools/probe/bw_probe ) > $OUT/${TAG}_bw_probe.txt 2>&1
( [ -x tools/probe/coissue_probe ] && timeout 120 tools/probe/coissue_probe ) > $OUT/${TAG}_coissue_probe.txt 2>&1
( [ -x tools/probe/mfma_peak_probe ] && timeout 120 tools/probe/mfma_peak_probe ) > $OUT/${TAG}_mfma_peak_probe.txt 2>&1
( [ -x tools/probe/bf16x3_probe ] && timeout 120 tools/probe/bf16x3_probe ) > $OUT/${TAG}_bf16x3_probe.txt 2>&1
# device idle gaps of the default bench step
bash tools/prof_gaps.sh > $OUT/${TAG}_gaps.txt 2>&1
cat $OUT/${TAG}_learner_rates.txt $OUT/${TAG}_shard_steps.txt
