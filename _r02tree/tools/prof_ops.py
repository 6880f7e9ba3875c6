#!/usr/bin/env python3
"""Host-side op counts of a learner update (torch profiler): how many launches / aten ops / copies one train() issues.
   python tools/prof_ops.py [alg] [envs]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
alg = sys.argv[1] if len(sys.argv) > 1 else "qplex"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 512
args = bench.make_args(alg, "2s3z", 0)
torch.manual_seed(0)
mac = SharedMAC(args); learner = QLearner(mac, args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args)
ep = w.generate_episodes(E)[0]
for i in range(4): learner.train(ep, i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for i in range(3): learner.train(ep, 10 + i)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:30]:
    print("%-60s count %4d cpu %8.1f us" % (e.key[:60], e.count, e.cpu_time_total))
