cd $GRAFT_REPO_ROOT
timeout 60 tools/probe/dma_probe > gpurun_out/r03_dma_probe.txt 2>&1; cat gpurun_out/r03_dma_probe.txt
( for r in 1 2; do for v in noil base; do
  if [ $v = base ]; then L=marl_amd/libmarl_hip.so; else L=marl_amd/variants/libmarl_hip_$v.so; fi
  MARL_HIP_LIB=$PWD/$L timeout 200 python tools/ktime.py --tag $v 2>&1 | grep -v amdgpu.ids
done; done
for v in noil base; do
  if [ $v = base ]; then L=marl_amd/libmarl_hip.so; else L=marl_amd/variants/libmarl_hip_$v.so; fi
  MARL_HIP_LIB=$PWD/$L timeout 200 python tools/ktime.py --tag $v --alg qplex --envs 4096 --rollouts 0 --updates 5 2>&1 | grep -v amdgpu.ids | head -8
done ) > gpurun_out/r03_ab4.txt 2>&1
cat gpurun_out/r03_ab4.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t5.log 2>&1; tail -3 gpurun_out/r03_t5.log
