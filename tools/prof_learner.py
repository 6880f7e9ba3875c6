#!/usr/bin/env python3
"""Profiling driver: N learner updates (and optionally rollouts) of the bench workload, nothing else.
Used under rocprofv3 (--kernel-trace / --pmc passes); prints nothing but a one-line summary."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--alg", default="qmix")
    ap.add_argument("--shape", default="2s3z")
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--updates", type=int, default=3)
    ap.add_argument("--rollouts", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=0, help="untimed updates before the timed ones (first-call allocations)")
    ap.add_argument("--mixer-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--gemm-mode", default="f32", choices=["f32", "bf16x6"])
    o = ap.parse_args()
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.algorithm.qtran_learner import QTRANLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    args = bench.make_args(o.alg, o.shape, o.T)
    args.mixer_dtype = o.mixer_dtype
    args.gemm_mode = o.gemm_mode
    torch.manual_seed(0)
    mac = SharedMAC(args)
    learner = QTRANLearner(mac, args) if o.alg.startswith("qtran") else QLearner(mac, args)
    env = SyntheticSMACEnv(o.envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit,
                           seed=1, fixed_length=True)
    w = RolloutWorker(env, mac, args)
    ep, _, _, _ = w.generate_episodes(o.envs)
    for _ in range(o.rollouts):
        w.generate_episodes(o.envs)
    for i in range(o.warmup):
        learner.train(ep, i)
    import gc
    gc.collect()
    gc.disable()            # a generation-2 collection costs 35-60 ms here - several updates (as bench.py / timeit do)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(o.updates):
        learner.train(ep, i)
    torch.cuda.synchronize()
    print("updates/s %.2f" % (o.updates / (time.perf_counter() - t0)))


if __name__ == "__main__":
    main()
