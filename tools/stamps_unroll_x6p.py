#!/usr/bin/env python3
"""Where a step of the round-6 plain unroll (csrc/agent_x6p.hip) goes, per wave of workgroup 0 (diagnostic build
`make -C marl_amd/csrc stamps`):  python tools/stamps_unroll_x6p.py [envs]
team R (waves 0-3): recurrence | X | (idle) Y       team I (waves 4-7): fc1 + q store | X | x + input planes + loads | Y"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MARL_HIP_LIB", os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so"))
sys.path.insert(0, HERE)
import torch  # noqa: E402
from marl_amd import _lib, ops  # noqa: E402
from stamps import show  # noqa: E402
import bench  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, O, S, A, T = bench.SHAPES["2s3z"]
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
fn = lib.marl_debug_stamps_agent_x6p
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf.data_ptr()) == 0
g = torch.Generator().manual_seed(0)
P = {"fc1.weight": torch.randn(64, O + A + N, generator=g) * 0.1, "fc1.bias": torch.randn(64, generator=g) * 0.1,
     "rnn.weight_ih": torch.randn(192, 64, generator=g) * 0.1, "rnn.weight_hh": torch.randn(192, 64, generator=g) * 0.1,
     "rnn.bias_ih": torch.randn(192, generator=g) * 0.1, "rnn.bias_hh": torch.randn(192, generator=g) * 0.1,
     "fc2.weight": torch.randn(A, 64, generator=g) * 0.1, "fc2.bias": torch.randn(A, generator=g) * 0.1}
w = ops.agent_weights({k: v.to(dev) for k, v in P.items()})
obs = torch.randn(B, T + 1, N, O, device=dev)
u = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32)
q = torch.empty(B, T, N, A, device=dev)
assert ops.agent_unroll_x6_plain_r6(B, T, N, O, A)
for _ in range(2):
    buf.zero_()
    ops.agent_unroll_fwd_x6(w, obs, (T + 1) * N, 1, u, T * N, 0, None, q, None, None, None, B, T, N, O, A)
    torch.cuda.synchronize()
show(buf.cpu().view(16, 16).numpy(), ["ph1", "X", "ph2", "Y"], "plain unroll, round-6 decomposition", B, T)
