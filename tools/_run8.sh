cd $GRAFT_REPO_ROOT
( for P in 0 1; do for D in 0 1; do MARL_NO_PAIR=$P MARL_FWD_W2L=$D timeout 200 python tools/ktime.py --tag nopair${P}_w2l$D --shape MMM2 --envs 1024 --rollouts 0 --mixer-dtype bf16 2>&1 | grep -v amdgpu.ids | head -7; done; done ) > gpurun_out/r03_ab7.txt 2>&1
cat gpurun_out/r03_ab7.txt
