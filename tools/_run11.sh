cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resident or qmix_wide" > gpurun_out/r03_t11.log 2>&1; tail -3 gpurun_out/r03_t11.log
( for r in 1 2; do for v in base ressb0; do
  if [ $v = base ]; then L=marl_amd/libmarl_hip.so; else L=marl_amd/variants/libmarl_hip_$v.so; fi
  rm -rf gpurun_out/r03_res_$v; MARL_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_res_$v -o p -- python3 tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 --warmup 3 --updates 10 --mixer-dtype bf16 > /dev/null 2>&1
  echo $v; grep -E "res_fwd" gpurun_out/r03_res_$v/p_kernel_stats.csv | cut -d, -f2-4
done; done ) 2>&1 | tee gpurun_out/r03_ab9.txt
