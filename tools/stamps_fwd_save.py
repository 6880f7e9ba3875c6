#!/usr/bin/env python3
"""Segment shares of the activation-saving unroll alone (diagnostic build), register prefetch vs LDS-DMA:
    python tools/stamps_fwd_save.py [shape] [envs]"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MARL_HIP_LIB"] = os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so")
sys.path.insert(0, HERE)
import numpy as np, torch  # noqa: E402
import bench  # noqa: E402
from marl_amd import _lib, ops  # noqa: E402
from stamps import show, SEGS  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "2s3z"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
args = bench.make_args("qmix", shape, 0)
N, O, A, T = args.n_agents, args.obs_shape, args.n_actions, args.episode_limit
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
fn = lib.marl_debug_stamps_fwd
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf.data_ptr()) == 0
from marl_amd.controller.share_params import SharedMAC
mac = SharedMAC(args); mac.cuda()
w = mac.agent.weights()
store = torch.randn(B, T + 1, N, O, device=dev)
u = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32)
q = torch.empty(B, T, N, A, device=dev); hl = torch.empty(B * N, 64, device=dev)
saved = torch.empty(ops.saved_shape(T, B, N), device=dev); gi = torch.empty(ops.saved_shape(T, B, N, planes=3), device=dev)
for mode in ("0", "1"):
    from marl_amd import experiments
    experiments.set("fwd_dma", int(mode))
    for _ in range(2):
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.agent_unroll_fwd(w, store, (T + 1) * N, 0, u, T * N, -1, None, q, None, hl, saved, B, T, N, O, A, gi_out=gi)
        e1.record(); torch.cuda.synchronize()
    print("MARL_FWD_DMA=%s : %.3f ms (stamped build)" % (mode, e0.elapsed_time(e1)))
    show(buf.cpu().view(16, 16).numpy(), SEGS["fwd"], "fwd save, DMA=" + mode, B, T)
