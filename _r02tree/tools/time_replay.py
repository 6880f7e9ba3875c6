#!/usr/bin/env python3
"""Times the replay legs of the bench step (store_episode / sample) on the device."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from marl_amd.controller.share_params import SharedMAC
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.common.replaybuffer import ReplayBuffer

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
args = bench.make_args("qmix", "2s3z", 0)
args.buffer_size = E
mac = SharedMAC(args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args)
buf = ReplayBuffer(args)
ep, _, _, _ = w.generate_episodes(E)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("store  ms", t(lambda: buf.store_episode(ep)))
print("sample ms", t(lambda: buf.sample(E)))
print("rollout ms", t(lambda: w.generate_episodes(E)))
