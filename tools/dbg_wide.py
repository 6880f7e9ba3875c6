import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from marl_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
E = 32
R, N, S = 1000, 5, 120
g = torch.Generator().manual_seed(R + N + S)
outs = {"w1": N * E, "b1": E, "w2": E, "h": E}
P = {}
for k in outs:
    P[k] = (torch.randn(outs[k], S, generator=g) * 0.2).requires_grad_()
    P[k + "_b"] = (torch.randn(outs[k], generator=g) * 0.2).requires_grad_()
P["b2_w"] = torch.randn(1, E, generator=g).requires_grad_()
P["b2_b"] = torch.randn(1, generator=g).requires_grad_()
s = torch.randn(R, S, generator=g)
q = torch.randn(R, N, generator=g, requires_grad=True)
gq = torch.randn(R, generator=g)
lin = lambda k: F.linear(s, P[k], P[k + "_b"])
w1 = lin("w1").abs().view(R, N, E)
pre = (q.unsqueeze(2) * w1).sum(1) + lin("b1")
hid = F.elu(pre)
qt = (hid * lin("w2").abs()).sum(1) + F.linear(torch.relu(lin("h")), P["b2_w"], P["b2_b"]).squeeze(1)
cu = lambda x: x.detach().float().to(dev).contiguous()
(qt * gq).sum().backward()
Wd = {k: cu(v) for k, v in P.items()}
base = {k: torch.randn(v.shape, generator=g) for k, v in P.items()}
Gd = {k: cu(v) for k, v in base.items()}
ld = (S + 3) // 4 * 4
sd = torch.zeros(R, ld, device=dev); sd[:, :S] = cu(s)
for rep in range(3):
    out = torch.full((R,), 9.0, device=dev)
    ops.qmix_wide_fwd(ops.qmix_weights(Wd), ops.src(sd[:, :S]), cu(q), out, R, N, S, E)
    o = out.cpu()
    bad = (~torch.isfinite(o)).nonzero().flatten().tolist()
    print("rep", rep, "nonfinite rows", bad, "maxdiff finite", float((o - qt.detach())[torch.isfinite(o)].abs().max()))
    if rep == 1:
        dq = torch.full((R, N), 9.0, device=dev)
        ops.qmix_wide_bwd(ops.qmix_weights(Wd), ops.src(sd[:, :S]), cu(q), cu(gq), dq, ops.qmix_weights(Gd), R, N, S, E)
        print("dq finite", bool(torch.isfinite(dq).all()), "maxdiff", float((dq.cpu() - q.grad).abs().max()))
        for k in P:
            d = (Gd[k].cpu() - base[k] - P[k].grad)
            if k == "w1":
                rowsum = d.abs().sum(1)
                c = int(rowsum.argmax()); print("  worst w1 column", c, "n=", c // E, "e=", c % E, "rowsum top5", rowsum.topk(5))
                # is the error one row's contribution?  d[c,:] ~ alpha * s[row,:]
                dc = d[c]
                proj = (s @ dc) / (s * s).sum(1)
                rr = int(proj.abs().argmax()); res = dc - proj[rr] * s[rr]
                print("  best single-row fit: row", rr, "alpha", float(proj[rr]), "residual", float(res.abs().max()), "of", float(dc.abs().max()))
                o_cpu = (F.linear(s, P["w1"], P["w1_b"]))[rr, c]
                print("  hypernet output at that (row, col):", float(o_cpu), " q", float(q[rr, c // E]))
            print(" grad", k, "finite", bool(torch.isfinite(Gd[k]).all()), "maxdiff", float(d.abs().max()), "scale", float(P[k].grad.abs().max()))
    for r in bad:
        print(" row", r, "ref", float(qt[r]), "pre min/max", float(pre[r].min()), float(pre[r].max()))
print("pre global max", float(pre.max()), "min", float(pre.min()))
