#!/usr/bin/env python3
"""Segment shares of the resident-weights bf16 QMIX forward (diagnostic build): python tools/stamps_wide.py [rows]"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MARL_HIP_LIB"] = os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so")
sys.path.insert(0, HERE)
import torch  # noqa: E402
from marl_amd import _lib, ops  # noqa: E402
from stamps import show  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 122880
N, S, E = 10, 322, 32
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
fn = lib.marl_debug_stamps_wide
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf.data_ptr()) == 0
g = torch.Generator().manual_seed(0)
W = {}
for k, n in (("w1", N * E), ("b1", E), ("w2", E), ("h", E)):
    W[k] = (torch.randn(n, S, generator=g) * 0.2).to(dev)
    W[k + "_b"] = (torch.randn(n, generator=g) * 0.2).to(dev)
W["b2_w"] = torch.randn(1, E, generator=g).to(dev); W["b2_b"] = torch.randn(1, generator=g).to(dev)
sd = torch.zeros(R, 324, device=dev); sd[:, :S] = torch.randn(R, S, device=dev)
q = torch.randn(R, N, device=dev); out = torch.empty(R, device=dev)
if len(sys.argv) > 2:
    from marl_amd import experiments
    experiments.set("wide_res32", int(sys.argv[2]))
for _ in range(3):
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.qmix_wide_fwd(ops.qmix_weights(W), ops.src(sd[:, :S]), q, out, R, N, S, E, bf16=True); e1.record()
    torch.cuda.synchronize()
print("call %.1f us (stamped build)" % (e0.elapsed_time(e1) * 1e3))
show(buf.cpu().view(16, 16).numpy(), ["top", "chunks", "qstage/mix", "mix/atomic", "prologue"], "qmix_wide resident forward", R, 1)
