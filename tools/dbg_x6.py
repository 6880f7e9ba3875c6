import sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from marl_amd import ops
dev = torch.device("cuda")
cu = lambda t, d, dt=None: t.to(d) if dt is None else t.to(d, dt)
rows, S, NH, HW, N3, G = int(sys.argv[1]), 120, 5, 11, 5, int(sys.argv[2])
g = torch.Generator().manual_seed(1)
K1 = S + NH * HW
x0 = torch.randn(rows, S, generator=g)
idx = torch.randint(-1, HW, (rows, NH), generator=g)
oh = torch.zeros(rows, NH, HW)
for j in range(NH):
    v = idx[:, j] >= 0
    oh[v, j, idx[v, j]] = 1
X = torch.cat([x0, oh.reshape(rows, -1)], 1)
sizes = [(64, K1), (64,), (64, 64), (64,), (N3, 64), (N3,)]
pad = lambda n: (n + 3) // 4 * 4
per = sum(pad(int(np.prod(z))) for z in sizes)
flat = torch.randn(G * per, generator=g) * 0.2
fd, gd = flat.to(dev), torch.zeros(G * per, device=dev)
def views(buf, k):
    out, off = [], k * per
    for z in sizes:
        n = int(np.prod(z)); out.append(buf[off:off + n].view(z)); off += pad(n)
    return out
class L:
    def __init__(self, w, b, gw, gb):
        self.weight, self.bias = torch.nn.Parameter(w, requires_grad=False), torch.nn.Parameter(b, requires_grad=False)
        self.weight.grad, self.bias.grad = gw, gb
heads = []
for k in range(G):
    w, gr = views(fd, k), views(gd, k)
    heads.append([L(w[2 * i], w[2 * i + 1], gr[2 * i], gr[2 * i + 1]) for i in range(3)])
xs = ops.src(x0.to(dev), idx=idx.to(dev, torch.int32), nhot=NH, hot_w=HW)
dY = torch.randn(rows, G * N3, generator=g)
outs = []
for rep in range(2):
    Y = torch.zeros(rows, G * N3, device=dev)
    hs = torch.full((ops.mlp3_save_floats(rows, True, G),), float("nan"), device=dev)
    ops.mlp3_fwd(ops.mlp3_weights(heads), xs, Y, rows, K1, N3, G, hsave=hs, x6=True)
    gd.zero_()
    ops.mlp3_bwd(ops.mlp3_weights(heads), xs, dY.to(dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G, hsave=hs, x6=True)
    torch.cuda.synchronize()
    outs.append(gd.cpu().clone())
print("run-to-run bitwise equal:", torch.equal(outs[0], outs[1]), "nan:", int(torch.isnan(outs[0]).sum()))
for k in range(G):
    ps = [v.double().clone().requires_grad_(True) for v in views(flat, k)]
    h = torch.relu(F.linear(X.double(), ps[0], ps[1])); h = torch.relu(F.linear(h, ps[2], ps[3])); y = F.linear(h, ps[4], ps[5])
    y.backward(dY[:, k * N3:(k + 1) * N3].double())
    line = "head %d:" % k
    for name, pr, gv in zip(("W1", "b1", "W2", "b2", "W3", "b3"), ps, views(outs[0], k)):
        d = (gv.double() - pr.grad).abs()
        sc = float(pr.grad.abs().max())
        line += " d%s %.1e" % (name, float(d.max()) / sc)
        if name == "W1" and float(d.max()) / sc > 1e-4:
            bad = (d / sc > 1e-4)
            print("   bad W1 cols:", sorted(set(bad.nonzero()[:, 1].tolist()))[:60], "rows:", sorted(set(bad.nonzero()[:, 0].tolist()))[:70])
    print(line)
