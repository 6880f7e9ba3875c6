#!/usr/bin/env python3
"""QTRAN head row-level kernels at the configs[3] shard size (3s5z, 512 episodes x 150 steps): the state parts and the
row-level weight gradients, new kernels vs the marl_linear / marl_linear_wgrad composition, same process.
    python tools/time_qtran.py [BT] [S]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import ops
from marl_amd.network.mixer import QtranQBase, QtranV
from marl_amd.hostutil import FlatParams

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 150
S = int(sys.argv[2]) if len(sys.argv) > 2 else 216
N, A = 8, 14
dev = torch.device("cuda:0")
args = types.SimpleNamespace(n_agents=N, n_actions=A, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
torch.manual_seed(0)
qn, vn = QtranQBase(args).to(dev), QtranV(args).to(dev)
fq, fv = FlatParams(list(qn.parameters()), dev, with_grad=True), FlatParams(list(vn.parameters()), dev, with_grad=True)
s = torch.randn(BT, S, device=dev)
h = torch.randn(BT * N, 64, device=dev) * 0.7
u = torch.randint(0, A, (BT * N,), device=dev).int()
d = torch.randn(BT, device=dev)
dh = torch.zeros(BT * N, 64, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


real_sp, real_wg = ops.qtran_state_parts_supported, ops.qtran_wgrad_rows_supported
for new in (False, True):
    ops.qtran_state_parts_supported = real_sp if new else (lambda S, s=None: False)
    ops.qtran_wgrad_rows_supported = real_wg if new else (lambda S, AE, s=None: False)
    t_sp1 = timed(lambda: qn.state_part(s, BT, "t"))
    t_sp2 = timed(lambda: qn.state_part(s, BT, "e", other=vn))
    cq, cv = {}, {}
    sp_e, sp_v = qn.state_part(s, BT, "e", other=vn)
    qn.hip_forward(s, h, u, BT, ctx=cq, tag="e", sp=sp_e)
    vn.hip_forward(s, h, BT, ctx=cv, sp=sp_v)
    t_bq = timed(lambda: qn.hip_backward(cq, d, BT, dh, accumulate=False))
    t_bv = timed(lambda: vn.hip_backward(cv, d, BT, dh, accumulate=True))
    print("%s  state part (1 head) %6.1f us   (Q + V heads) %6.1f us   backward Q %6.1f us   backward V %6.1f us" %
          ("new kernels " if new else "composition ", t_sp1, t_sp2, t_bq, t_bv))
