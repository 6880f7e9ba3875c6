#!/usr/bin/env python3
"""Per-update wall time of one bench leg (bench.config_leg's setup), to see what falls into its first timed segment.
usage: tools/leg_updates.py [alg] [shape] [envs] [gemm_mode] [updates]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.algorithm.qtran_learner import QTRANLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.common.replaybuffer import ReplayBuffer
from marl_amd.env.synthetic_smac import SyntheticSMACEnv

alg = sys.argv[1] if len(sys.argv) > 1 else "qplex"
shape = sys.argv[2] if len(sys.argv) > 2 else "2s3z"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 512
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x6"
n = int(sys.argv[5]) if len(sys.argv) > 5 else 48
args = bench.make_args(alg, shape, 0)
args.gemm_mode = mode
args.buffer_size, args.batch_size = 2 * envs, envs
torch.manual_seed(0); np.random.seed(1)
mac = SharedMAC(args)
learner = QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args)
env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
worker = RolloutWorker(env, mac, args)
buf = ReplayBuffer(args)
worker.record_sink = buf
for _ in range(2):
    buf.store_episode(worker.generate_episodes(envs)[0])
gc.collect(); gc.disable()
ts = []
for i in range(n):
    t0 = time.perf_counter()
    learner.train(buf.sample(envs), i)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("%s %s %d envs %s: per-update ms (synchronised each)" % (alg, shape, envs, mode))
for k in range(0, n, 12):
    print("  updates %2d-%2d: %s" % (k, k + 11, " ".join("%6.2f" % x for x in ts[k:k + 12])))
