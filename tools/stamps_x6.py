#!/usr/bin/env python3
"""Segment shares of the bf16x6 head backward (diagnostic build `make -C marl_amd/csrc stamps`): python tools/stamps_x6.py [envs]"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MARL_HIP_LIB"] = os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so")
sys.path.insert(0, HERE)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from marl_amd import _lib, ops  # noqa: E402
from stamps import show  # noqa: E402
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows, G, S, K1, N3 = envs * 120, 10, 120, 120, 5
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
fn = lib.marl_debug_stamps_mlp3x6
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf.data_ptr()) == 0
s = torch.randn(rows, S, device=dev)
x = ops.src(s)
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
Y = torch.empty(rows, G * N3, device=dev); dY = torch.randn(rows, G * N3, device=dev)
w, gw = ops.mlp3_weights(heads), ops.mlp3_weights(heads, grad=True)
hs = torch.empty(ops.mlp3_save_floats(rows, True, G), device=dev)
ops.mlp3_fwd(w, x, Y, rows, K1, N3, G, hsave=hs, x6=True)
for _ in range(2):
    buf.zero_()
    ops.mlp3_bwd(w, x, dY, gw, rows, K1, N3, G, hsave=hs, x6=True)
    torch.cuda.synchronize()
its = (rows + 63) // 64 // 48
show(buf.cpu().view(16, 16).numpy(), ["phaseA", "put h1", "bar", "dW2", "bar", "put h2", "bar", "dW3", "bar", "issue_x", "bar", "dW1+bar", "put dh1", "put x", "ones"], "mlp3 x6 backward", envs, its)
