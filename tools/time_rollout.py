#!/usr/bin/env python3
"""Whole-rollout kernels: fp32 MFMA (csrc/rollout_fused.hip) vs the bf16x6 split kernels (csrc/rollout_x6.hip, and its round-5 twin
csrc/rollout_x6_v1.hip behind the rollout_v1 experiment switch), 2s3z shape, T = 120.
    [SHAPE=2s3z|3s5z] python tools/time_rollout.py [envs ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd import experiments
shape = os.environ.get("SHAPE", "2s3z")
for envs in [int(x) for x in sys.argv[1:]] or [4096, 2048, 1024, 512]:
    recs = {}
    for mode in ("f32", "x6_v1", "x6_r6", "bf16x6"):
        args = bench.make_args("qmix", shape, 0)
        args.gemm_mode = "f32" if mode == "f32" else "bf16x6"
        experiments.set("rollout_v1", {"x6_v1": 1, "x6_r6": 2}.get(mode, 0))      # forced round-5 / round-6 kernel; bf16x6 = the library's choice
        torch.manual_seed(0)
        mac = SharedMAC(args); mac.cuda()
        env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1)
        w = RolloutWorker(env, mac, args)
        w.epsilon = 0.3
        ep = w.generate_episodes(envs)[0]
        recs[mode] = ep.record.u.clone()
        for _ in range(2): w.generate_episodes(envs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        steps = 0
        for _ in range(5): steps += w.generate_episodes(envs)[3]
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("envs %5d  %-7s %.3f ms per rollout  %.1f M env-steps/s" % (envs, mode, ms, steps / 5 / ms / 1e3))
    d = int((recs["f32"] != recs["bf16x6"]).flatten(1).any(1).sum().item())
    print("           episodes whose actions differ between the two arithmetic modes: %d of %d" % (d, envs))
