#!/usr/bin/env python3
"""Times the fused three-layer head kernels (marl_mlp3_fwd / marl_mlp3_bwd) on the QPLEX lambda-net shapes:
rows = envs * T, 10 heads, x = [state 120] or [state 120 | one-hot 5 x 11].  Prints ms and fp32 TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import ops
import torch.nn as nn

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows, S, N, A, G = envs * 120, 120, 5, 11, 10
dev = torch.device("cuda:0")
s = torch.randn(rows, S, device=dev)
u = torch.randint(0, A, (rows, N), device=dev, dtype=torch.int32)
for name, K1, N3, x in (("key", S, 1, ops.src(s)), ("agents", S, N, ops.src(s)),
                        ("action", S + N * A, N, ops.src(s, idx=u, nhot=N, hot_w=A))):
    sizes = [(64, K1), (64,), (64, 64), (64,), (N3, 64), (N3,)]
    pad = lambda n: (n + 3) // 4 * 4
    per = sum(pad(torch.Size(z).numel()) for z in sizes)
    flat, grad = torch.randn(G * per, device=dev) * 0.1, torch.zeros(G * per, device=dev)
    heads = []
    for k in range(G):
        off, ls = k * per, []
        for li in range(3):
            l = nn.Linear(1, 1)
            for attr, z in (("weight", sizes[2 * li]), ("bias", sizes[2 * li + 1])):
                n = torch.Size(z).numel()
                p = nn.Parameter(flat[off:off + n].view(z), requires_grad=False)
                p.grad = grad[off:off + n].view(z)
                setattr(l, attr, p)
                off += pad(n)
            ls.append(l)
        heads.append(ls)
    Y = torch.empty(rows, G * N3, device=dev)
    dY = torch.randn(rows, G * N3, device=dev)
    w, gw = ops.mlp3_weights(heads), ops.mlp3_weights(heads, grad=True)
    fl = 2.0 * rows * G * (K1 * 64 + 64 * 64 + 64 * N3)
    for what, fn, mult in (("fwd", lambda: ops.mlp3_fwd(w, x, Y, rows, K1, N3, G), 1.0),
                           ("bwd", lambda: ops.mlp3_bwd(w, x, dY, gw, rows, K1, N3, G), 3.0 - (K1 * 64) / (K1 * 64 + 64 * 64 + 64 * N3))):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%-7s %s  K1=%3d N3=%d  %.3f ms  %.1f TFLOP/s (algorithmic %.1f GFLOP)" % (name, what, K1, N3, ms, fl * mult / ms / 1e9, fl * mult / 1e9))
