#!/usr/bin/env python3
"""Host-side cost of one C-ABI launch (no sync inside the loop): a trivial kernel, a small torch op and marl_linear.
Run it a few times: on the two-socket GPU box the per-launch cost depends on which NUMA node the process lands on.
    python tools/launch_cost.py [--pin]      --pin: bind to the CPUs of the GPU's NUMA node first (marl_amd.hostutil.pin_to_gpu_numa)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if "--pin" in sys.argv:
    from marl_amd.hostutil import pin_to_gpu_numa
    print("pinned:", pin_to_gpu_numa(0))
from marl_amd import ops
dev = torch.device("cuda:0")
a = torch.zeros(256, device=dev); b = torch.zeros(256, device=dev); c = torch.zeros(256, device=dev)
x = torch.randn(64, 120, device=dev); Y = torch.empty(64, 20, device=dev)
def timeit(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("vec_add launch        %.1f us" % timeit(lambda: ops.vec_add(a, b, c, 256)))
print("torch add_            %.1f us" % timeit(lambda: a.add_(1.0)))
W = torch.randn(20, 120, device=dev)
print("marl_linear launch    %.1f us" % timeit(lambda: ops.linear(ops.src(x), W, None, Y, 64, 20, 120)))
aff = sorted(os.sched_getaffinity(0))
print("affinity", aff[:4], "...", aff[-2:], "n =", len(aff), "cpu now", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
