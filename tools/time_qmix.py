#!/usr/bin/env python3
"""Fused QMIX mixer kernels (csrc/qmix_fused.hip), fp32 MFMA vs the bf16x6 split variant: forward, backward, loss + backward.
    python tools/time_qmix.py [envs ...]      (2s3z shape: N = 5, S = 120, T = 120)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import ops
dev = torch.device("cuda:0")
N, S, E, T = 5, 120, 32, 120
g = torch.Generator().manual_seed(0)
W = {}
for k, n in (("w1", N * E), ("b1", E), ("w2", E), ("h", E)):
    W[k] = (torch.randn(n, S, generator=g) * 0.2).to(dev)
    W[k + "_b"] = (torch.randn(n, generator=g) * 0.2).to(dev)
W["b2_w"] = torch.randn(1, E, generator=g).to(dev); W["b2_b"] = torch.randn(1, generator=g).to(dev)
G = {k: torch.zeros_like(v) for k, v in W.items()}
for envs in [int(x) for x in sys.argv[1:]] or [4096, 1024, 512]:
    R = envs * T
    s = torch.randn(R, S, device=dev); q = torch.randn(R, N, device=dev); gq = torch.randn(R, device=dev)
    out = torch.empty(R, device=dev); dq = torch.empty(R, N, device=dev)
    r, term, pad, tgt = torch.randn(R, device=dev), torch.zeros(R, device=dev), torch.zeros(R, device=dev), torch.randn(R, device=dev)
    loss2 = torch.zeros(2, device=dev)
    xs = ops.src(s)
    flop = 2.0 * R * S * (N * E + 3 * E)
    for x6 in (False, True):
        legs = [("forward        ", 1, lambda: ops.qmix_fused_fwd(ops.qmix_weights(W), xs, q, out, R, N, S, E, x6=x6)),
                ("backward       ", 2, lambda: ops.qmix_fused_bwd(ops.qmix_weights(W), xs, q, gq, dq, ops.qmix_weights(G), R, N, S, E, x6=x6)),
                ("loss + backward", 2, lambda: ops.qmix_fused_loss_bwd(ops.qmix_weights(W), xs, q, tgt, r, term, pad, 0.99, out, dq, ops.qmix_weights(G), loss2, R, N, S, E, x6=x6))]
        for name, mult, fn in legs:
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("envs %5d  %s %s %.3f ms  %.1f TFLOP/s of fp32 work" % (envs, "bf16x6   " if x6 else "fp32 MFMA", name, ms, mult * flop / ms / 1e9))
