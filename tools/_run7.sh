cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "variants or config5" > gpurun_out/r03_t7.log 2>&1; tail -4 gpurun_out/r03_t7.log
( for D in 0 1; do MARL_FWD_W2L=$D timeout 200 python tools/ktime.py --tag w2l$D --shape MMM2 --envs 1024 --rollouts 0 --mixer-dtype bf16 2>&1 | grep -v amdgpu.ids | head -5; done ) > gpurun_out/r03_ab6.txt 2>&1
cat gpurun_out/r03_ab6.txt
