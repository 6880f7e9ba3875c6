#!/usr/bin/env python3
"""Host-side cost of one learner update (small shards are bound by it): cProfile of N train() calls."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    args = bench.make_args("qmix", "2s3z", 0)
    torch.manual_seed(0)
    mac = SharedMAC(args)
    learner = QLearner(mac, args)
    env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
    w = RolloutWorker(env, mac, args)
    ep = w.generate_episodes(envs)[0]
    for i in range(5):
        learner.train(ep, i)
    import gc
    gc.collect(); gc.disable()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for i in range(n):
        learner.train(ep, i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("per update: host enqueue %.3f ms, host+device %.3f ms" % (t_host / n * 1e3, t_all / n * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        learner.train(ep, i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
