#!/usr/bin/env python3
"""Times learner.train on a replay sample view (ring + index) vs a plain record."""
import os, sys, time
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.common.replaybuffer import ReplayBuffer
E = 4096
args = bench.make_args("qmix", "2s3z", 0); args.buffer_size = 2 * E
mac = SharedMAC(args); learner = QLearner(mac, args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args); buf = ReplayBuffer(args); w.record_sink = buf
for _ in range(2):
    ep = w.generate_episodes(E)[0]; buf.store_episode(ep)
batch = buf.sample(E)
for name, b in (("view", batch), ("record", ep), ("view", batch), ("sorted-view", None)):
    if b is None:
        b = buf.sample(E); b.index = torch.sort(b.index)[0]
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        learner.train(b, 1 + i)
        torch.cuda.synchronize(); print(name, i, "%.2f ms" % ((time.perf_counter() - t0) * 1e3))
