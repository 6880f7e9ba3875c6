cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t4.log 2>&1; tail -12 gpurun_out/r03_t4.log
( timeout 100 python tools/stamps_fwd_save.py 2s3z 4096; timeout 100 python tools/stamps_fwd_save.py MMM2 1024 ) 2>&1 | grep -v amdgpu > gpurun_out/r03_stamps_dma.txt; cat gpurun_out/r03_stamps_dma.txt
( for D in 0 -1; do MARL_FWD_DMA=$D timeout 200 python tools/ktime.py --tag dma$D --shape MMM2 --envs 1024 --rollouts 0 --mixer-dtype bf16 2>&1 | grep -v amdgpu.ids | head -12; done ) > gpurun_out/r03_ab3.txt 2>&1; cat gpurun_out/r03_ab3.txt
