import sys, os, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_kernels as t
from marl_amd import ops
dev = torch.device("cuda:0")
orig = ops.qmix_wide_fwd
def spy(w, s, q, out, rows, N, S, E, bf16=False):
    orig(w, s, q, out, rows, N, S, E, bf16=bf16)
    torch.cuda.synchronize()
    bad = (~torch.isfinite(out)).nonzero().flatten().tolist()
    print("fwd nonfinite rows:", bad, "src ld0", s.ld0, "k0", s.k0, "p0 %16 =", s.p0 % 16, "q ptr%16", q.data_ptr() % 16, "q shape", tuple(q.shape), q.stride())
    out2 = torch.full_like(out, 7.0)
    orig(w, s, q, out2, rows, N, S, E, bf16=bf16)
    print("second call nonfinite:", (~torch.isfinite(out2)).nonzero().flatten().tolist())
ops.qmix_wide_fwd = spy
try:
    t.test_qmix_wide(dev, 1000, 5, 120, False)
    print("PASSED")
except AssertionError as e:
    print("FAILED", str(e)[:300])
