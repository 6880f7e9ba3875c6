cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t3.log 2>&1; tail -15 gpurun_out/r03_t3.log
( for D in 0 1; do MARL_FWD_DMA=$D timeout 200 python tools/ktime.py --tag dma$D --rollouts 0 2>&1 | grep -v amdgpu.ids | head -6; done
  for D in 0 -1; do MARL_FWD_DMA=$D timeout 200 python tools/ktime.py --tag dma$D --shape MMM2 --envs 1024 --rollouts 0 2>&1 | grep -v amdgpu.ids | head -7; done ) > gpurun_out/r03_ab2.txt 2>&1
cat gpurun_out/r03_ab2.txt
timeout 200 python tools/stamps.py mlp3 4096 > gpurun_out/r03_stamps_mlp3.txt 2>&1; cat gpurun_out/r03_stamps_mlp3.txt | grep -v amdgpu
