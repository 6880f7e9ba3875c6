#!/usr/bin/env python3
"""Where a step of the split unroll (csrc/agent_x6.hip) spends its cycles, per wave of workgroup 0 (diagnostic build
`make -C marl_amd/csrc stamps`):  python tools/stamps_unroll_x6.py [envs] [plain|save|read|bwd]
team R (waves 0-3): h W_hh products | gate math + h planes | stores + fc2 | barrier;  team I (waves 4-7): gi | fc1 | input tile | barrier
mode bwd (csrc/agent_bwd_x6.hip): team R: gate gradients + image | barrier | carry chain | dW_hh;  team I: barrier | dx | dW_ih | dW_2 + hand-offs"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MARL_HIP_LIB", os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so"))
sys.path.insert(0, HERE)
import torch  # noqa: E402
from marl_amd import _lib, ops  # noqa: E402
from stamps import show  # noqa: E402
import bench  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
for fn in (lib.marl_debug_stamps_agent_x6, lib.marl_debug_stamps_agent_bwd_x6):
    fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
    assert fn(buf.data_ptr()) == 0
N, O, S, A, T = bench.SHAPES[os.environ.get("SHAPE", "2s3z")]
g = torch.Generator().manual_seed(0)
P = {"fc1.weight": torch.randn(64, O + A + N, generator=g) * 0.1, "fc1.bias": torch.randn(64, generator=g) * 0.1,
     "rnn.weight_ih": torch.randn(192, 64, generator=g) * 0.1, "rnn.weight_hh": torch.randn(192, 64, generator=g) * 0.1,
     "rnn.bias_ih": torch.randn(192, generator=g) * 0.1, "rnn.bias_hh": torch.randn(192, generator=g) * 0.1,
     "fc2.weight": torch.randn(A, 64, generator=g) * 0.1, "fc2.bias": torch.randn(A, generator=g) * 0.1}
w = ops.agent_weights({k: v.to(dev) for k, v in P.items()})
obs = torch.randn(B, T + 1, N, O, device=dev)
u = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32)
q = torch.empty(B, T, N, A, device=dev)
saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
gi = torch.empty(ops.saved_shape(T, B, N, planes=3), device=dev)
hl = torch.zeros(B * N, 64, device=dev)
cus = int(os.environ.get("CUS", "0"))
x6 = ops.agent_unroll_fwd_x6
x6(w, obs, (T + 1) * N, 0, u, T * N, -1, None, q, None, hl, saved, B, T, N, O, A, cu_budget=cus, gi_out=gi)
for _ in range(2):
    buf.zero_()
    if mode == "plain":
        x6(w, obs, (T + 1) * N, 1, u, T * N, 0, None, q, None, None, None, B, T, N, O, A, cu_budget=cus)
    elif mode == "bwd":
        pd = {k: v.to(dev) for k, v in P.items()}
        names = ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh", "fc2.weight", "fc2.bias")
        grads = {k: torch.zeros_like(pd[k]) for k in names}
        dxp = torch.empty(B, T, N, 64, device=dev)
        di, dv = torch.randint(0, A, (B, T, N), device=dev, dtype=torch.int32), torch.randn(B, T, N, device=dev)
        ops.agent_unroll_bwd(w, None, None, saved, None, dxp, None, grads, B, T, N, A, dq_idx=di, dq_val=dv, x6=True)
    elif mode == "save":
        x6(w, obs, (T + 1) * N, 0, u, T * N, -1, None, q, None, hl, saved, B, T, N, O, A, cu_budget=cus, gi_out=gi)
    else:
        x6(w, obs, (T + 1) * N, 1, u, T * N, 0, hl, q, None, None, None, B, T, N, O, A, cu_budget=cus, gi_in=gi)
    torch.cuda.synchronize()
show(buf.cpu().view(16, 16).numpy(), ["seg0", "seg1", "seg2", "barrier"], "split unroll, " + mode, B, T)
