cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_learners.py -m gpu -x -q -k "wide or config5 or MMM2 or folded" > gpurun_out/r03_t9.log 2>&1; tail -4 gpurun_out/r03_t9.log
( for D in 0 1; do MARL_WIDE_RES=$D timeout 200 python tools/ktime.py --tag res$D --shape MMM2 --envs 1024 --rollouts 0 --mixer-dtype bf16 2>&1 | grep -v amdgpu.ids | grep -E "==|qmix_wide"; done ) > gpurun_out/r03_ab8.txt 2>&1
cat gpurun_out/r03_ab8.txt
