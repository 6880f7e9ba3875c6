#!/usr/bin/env python3
"""Per-kernel HIP-event times of the learner update and the rollout at a given shape (bench.KernelTimers), one line per
kernel: for A/B runs of library variants in ONE gpurun call (MARL_HIP_LIB=<variant .so> python tools/ktime.py ...).
    python tools/ktime.py [--alg qmix] [--shape 2s3z] [--envs 4096] [--updates 10] [--rollouts 5] [--mixer-dtype fp32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--alg", default="qmix")
    ap.add_argument("--shape", default="2s3z")
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--rollouts", type=int, default=5)
    ap.add_argument("--mixer-dtype", default="fp32")
    ap.add_argument("--tag", default="")
    ap.add_argument("--two-hyper", action="store_true", help="QMIX with two_hyper_layers=True (network/mixer.py:36-43)")
    ap.add_argument("--gemm-mode", default="f32", choices=["f32", "bf16x6"])
    o = ap.parse_args()
    from marl_amd import ops, _lib
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.algorithm.qtran_learner import QTRANLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    args = bench.make_args(o.alg, o.shape, 0)
    args.mixer_dtype = o.mixer_dtype
    args.gemm_mode = o.gemm_mode
    if o.two_hyper:
        args.two_hyper_layers = True
    torch.manual_seed(0)
    mac = SharedMAC(args)
    learner = QTRANLearner(mac, args) if o.alg.startswith("qtran") else QLearner(mac, args)
    env = SyntheticSMACEnv(o.envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit,
                           seed=1, fixed_length=True)
    w = RolloutWorker(env, mac, args)
    timers = bench.KernelTimers(ops, args, o.envs)
    ep = w.generate_episodes(o.envs)[0]
    for i in range(3):
        learner.train(ep, i)
    torch.cuda.synchronize()
    import gc, time
    gc.collect(); gc.disable()
    timers.on = True
    for _ in range(o.rollouts):
        w.generate_episodes(o.envs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(o.updates):
        learner.train(ep, 3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(o.updates, 1)
    timers.on = False
    print("== %s %s %s envs=%d lib=%s : %.3f ms per update (%.1f updates/s)" % (o.tag, o.alg, o.shape, o.envs,
          os.path.basename(_lib.LIB_PATH), dt * 1e3, 1.0 / dt if dt else 0))
    for e in timers.table():
        ms = [a.elapsed_time(b) for a, b in timers.rec[e["name"]]["ev"]]
        print("  %-92s n=%3d  mean %.4f  median %.4f  min %.4f ms  %.1f TF exec (%.3f)" % (e["name"][:92], len(ms), e["ms"], float(np.median(ms)),
              min(ms), e["tflops"], e["frac"]))


if __name__ == "__main__":
    main()
