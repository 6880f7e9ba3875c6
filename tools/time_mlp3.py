#!/usr/bin/env python3
"""Times the fused three-layer head kernels (marl_mlp3_fwd / marl_mlp3_bwd) on the QPLEX lambda-net shapes:
rows = envs * T, 10 heads, x = [state 120] or [state 120 | one-hot 5 x 11].  Prints ms and fp32 TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import ops
import torch.nn as nn

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape = sys.argv[2] if len(sys.argv) > 2 else "2s3z"          # MMM2: K1 = 322 / 502 (kept activations only)
S, N, A = {"2s3z": (120, 5, 11), "3s5z": (216, 8, 14), "MMM2": (322, 10, 18)}[shape]
rows, G = envs * 120, 10
dev = torch.device("cuda:0")
s = torch.randn(rows, (S + 3) // 4 * 4, device=dev)[:, :S]      # rows padded to 16 bytes, as the episode record stores them
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
    hs = torch.empty(ops.mlp3_save_floats(rows, True, G), device=dev)
    bm = 3.0 - (K1 * 64) / (K1 * 64 + 64 * 64 + 64 * N3)
    legs = [("fwd", lambda: ops.mlp3_fwd(w, x, Y, rows, K1, N3, G), 1.0),
            ("bwd", lambda: ops.mlp3_bwd(w, x, dY, gw, rows, K1, N3, G), bm),
            ("fwd keeping h1,h2", lambda: ops.mlp3_fwd(w, x, Y, rows, K1, N3, G, hsave=hs), 1.0),
            ("bwd from kept h1,h2", lambda: ops.mlp3_bwd(w, x, dY, gw, rows, K1, N3, G, hsave=hs), bm)]
    if ops.mlp3_needs_kept(x, K1):
        del legs[1]
    if ops.mlp3_x6_supported(x, K1, 64, 64, N3, G):      # the bf16x6 split pair (csrc/mlp3_x6.hip)
        hs6 = torch.empty(ops.mlp3_save_floats(rows, True, G), device=dev)
        legs += [("x6 fwd", lambda: ops.mlp3_fwd(w, x, Y, rows, K1, N3, G, x6=True), 1.0),
                 ("x6 fwd keeping", lambda: ops.mlp3_fwd(w, x, Y, rows, K1, N3, G, hsave=hs6, x6=True), 1.0),
                 ("x6 bwd from kept", lambda: ops.mlp3_bwd(w, x, dY, gw, rows, K1, N3, G, hsave=hs6, x6=True), bm)]
    for what, fn, mult in legs:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%-7s %-19s  K1=%3d N3=%d  %.3f ms  %.1f TFLOP/s (algorithmic %.1f GFLOP)" % (name, what, K1, N3, ms, fl * mult / ms / 1e9, fl * mult / 1e9))
