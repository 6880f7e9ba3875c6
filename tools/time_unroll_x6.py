#!/usr/bin/env python3
"""Agent unroll (plain / activation-saving / gate-sum-reading): fp32 MFMA kernel (agent.hip) vs the bf16x6 split kernel (agent_x6.hip), 2s3z-sized agent, T = 120.
    [CUS=128] [SHAPE=2s3z|3s5z|MMM2] python tools/time_unroll_x6.py [envs ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import ops
import bench
dev = torch.device("cuda:0")
SHAPE = os.environ.get("SHAPE", "2s3z")
N, O, S, A, T = bench.SHAPES[SHAPE]
g = torch.Generator().manual_seed(0)
P = {"fc1.weight": torch.randn(64, O + A + N, generator=g) * 0.1, "fc1.bias": torch.randn(64, generator=g) * 0.1,
     "rnn.weight_ih": torch.randn(192, 64, generator=g) * 0.1, "rnn.weight_hh": torch.randn(192, 64, generator=g) * 0.1,
     "rnn.bias_ih": torch.randn(192, generator=g) * 0.1, "rnn.bias_hh": torch.randn(192, generator=g) * 0.1,
     "fc2.weight": torch.randn(A, 64, generator=g) * 0.1, "fc2.bias": torch.randn(A, generator=g) * 0.1}
w = ops.agent_weights({k: v.to(dev) for k, v in P.items()})
Fa = 2 * (O + A + N) * 64 + 12 * 64 * 64 + 2 * 64 * A
for B in [int(x) for x in sys.argv[1:]] or [4096, 1024, 512]:
    obs = torch.randn(B, T + 1, N, O, device=dev)
    u = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32)
    q, q6 = torch.empty(B, T, N, A, device=dev), torch.empty(B, T, N, A, device=dev)
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
    gi, gi6 = torch.empty(ops.saved_shape(T, B, N, planes=3), device=dev), torch.empty(ops.saved_shape(T, B, N, planes=3), device=dev)
    hl = torch.zeros(B * N, 64, device=dev)
    cus = int(os.environ.get("CUS", "0"))
    f32, x6 = ops.agent_unroll_fwd, ops.agent_unroll_fwd_x6
    from marl_amd import experiments
    def x6_r5(*a_, **k_):            # the round-5 plain kernel (csrc/agent_x6.hip) where the library would pick csrc/agent_x6p.hip
        with experiments.override(unroll_r6=0):
            x6(*a_, **k_)
    legs = [("fp32 MFMA        ", lambda: f32(w, obs, (T + 1) * N, 1, u, T * N, 0, None, q, None, None, None, B, T, N, O, A, cu_budget=cus)),
            ("bf16x6 (agent_x6)", lambda: x6_r5(w, obs, (T + 1) * N, 1, u, T * N, 0, None, q6, None, None, None, B, T, N, O, A, cu_budget=cus)),
            ("bf16x6           ", lambda: x6(w, obs, (T + 1) * N, 1, u, T * N, 0, None, q6, None, None, None, B, T, N, O, A, cu_budget=cus)),
            ("bf16x6 saving -gi", lambda: x6(w, obs, (T + 1) * N, 0, u, T * N, -1, None, q6, None, hl, saved, B, T, N, O, A, cu_budget=cus)),
            ("fp32 MFMA saving ", lambda: f32(w, obs, (T + 1) * N, 0, u, T * N, -1, None, q, None, hl, saved, B, T, N, O, A, cu_budget=cus, gi_out=gi)),
            ("bf16x6 saving    ", lambda: x6(w, obs, (T + 1) * N, 0, u, T * N, -1, None, q6, None, hl, saved, B, T, N, O, A, cu_budget=cus, gi_out=gi6)),
            ("fp32 MFMA reading", lambda: f32(w, obs, (T + 1) * N, 1, u, T * N, 0, hl, q, None, None, None, B, T, N, O, A, cu_budget=cus, gi_in=gi)),
            ("bf16x6 reading   ", lambda: x6(w, obs, (T + 1) * N, 1, u, T * N, 0, hl, q6, None, None, None, B, T, N, O, A, cu_budget=cus, gi_in=gi6))]
    hs = None
    pd = {k: v.to(dev) for k, v in P.items()}
    names = ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh", "fc2.weight", "fc2.bias")
    grads = {k: torch.zeros_like(pd[k]) for k in names}
    dxp = torch.empty(B, T, N, 64, device=dev)
    di, dv = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32), torch.randn(B, T, N, device=dev)
    bw = lambda x6_: ops.agent_unroll_bwd(w, None, None, saved, hs, dxp, None, grads, B, T, N, A, dq_idx=di, dq_val=dv, x6=x6_)
    if os.environ.get("BWD", "1") == "1":
        legs += [("fp32 MFMA BPTT   ", lambda: bw(False)), ("bf16x6 BPTT      ", lambda: bw(True))]
    for name, fn in legs:
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("envs %5d  %s %.3f ms  %.1f TFLOP/s (fp32 work %.1f GFLOP)  %.2f us per step" % (B, name, ms, Fa * B * N * T / ms / 1e9, Fa * B * N * T / 1e9, ms * 1e3 / T))
    print("           max |q6 - q| = %.2e (scale %.2e)" % (float((q6 - q).abs().max()), float(q.abs().max())))
