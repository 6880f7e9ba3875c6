"""Kernel-level parity: every C-ABI entry point vs the CPU oracle / plain torch-CPU fp32.
Needs a real MI355X: ``pytest -m gpu``.  Tolerances: 1e-4 absolute on O(1) fp32 values
(north_star), tighter where the arithmetic is short."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import seeded, nets, rollout as orl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from marl_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def cu(x, dev, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to(dtype).to(dev).contiguous()


def close(a, b, atol=1e-4, rtol=1e-4, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=atol, rtol=rtol, err_msg=msg)


# ------------------------------------------------------------------------------------- dense
@pytest.mark.parametrize("M,N,K,act", [(300, 256, 120, 0), (37, 11, 64, 1), (129, 1, 33, 0), (16, 70, 7, 1),
                                       (2050, 64, 176, 1)])
def test_linear_fwd(dev, M, N, K, act):
    from marl_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    X, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2, torch.randn(N, generator=g)
    ref = F.linear(X, W, b)
    if act:
        ref = torch.relu(ref)
    Xd, Wd, bd = cu(X, dev), cu(W, dev), cu(b, dev)
    Y = torch.full((M, N), 7.0, device=dev)
    ops.linear(ops.src(Xd), Wd, bd, Y, M, N, K, act=act)
    close(Y, ref, 2e-4)
    # dX = dY W through the k-major path, with relu gate and accumulate
    Yact = torch.randn(M, N, generator=g)
    dY = torch.randn(M, N, generator=g)
    base = torch.randn(M, K, generator=g)
    refdx = base + (dY * (Yact > 0)) @ W
    dX = cu(base, dev)
    ops.linear(ops.src(cu(dY, dev), gate=cu(Yact, dev)), Wd, None, dX, M, K, N, beta=1.0, w_kmajor=True)
    close(dX, refdx, 3e-4)


@pytest.mark.parametrize("M,N,K", [(300, 416, 322), (1000, 64, 120)])
def test_linear_bf16_operands(dev, M, N, K):
    """opt-in bf16 mixer GEMMs (BASELINE config 5): operands rounded to bf16 (RNE), fp32 accumulation - compared
    with the same rounding in torch (tight) and with exact fp32 (tolerance 2e-2 of the output scale)."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(M + K)
    X, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    Y = torch.empty(M, N, device=dev)
    ops.linear(ops.src(cu(X, dev)), cu(W, dev), cu(b, dev), Y, M, N, K, bf16=True)
    close(Y, F.linear(rb(X), rb(W), b), 1e-3, 1e-3)
    exact = F.linear(X, W, b)
    assert float((Y.cpu() - exact).abs().max()) < 2e-2 * float(exact.abs().max())
    dY = torch.randn(M, N, generator=g)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_wgrad(cu(dY, dev), ops.src(cu(X, dev)), dW, db, M, N, K, bf16=True)
    ref = rb(dY).t() @ rb(X)
    close(dW / M ** 0.5, ref / M ** 0.5, 2e-3, 2e-3)
    dX = torch.empty(M, K, device=dev)
    ops.linear(ops.src(cu(dY, dev)), cu(W, dev), None, dX, M, K, N, w_kmajor=True, bf16=True)
    close(dX, rb(dY) @ rb(W), 2e-3, 2e-3)


def test_linear_and_wgrad_random_shapes(dev):
    """Seeded sweep over odd sizes and virtual-concat compositions (aligned / unaligned dense segment 0, second
    dense segment, one-hot blocks, agent id, relu gate, k-major weights): every dispatch branch of the GEMM kernels
    (16-byte / dword / element operand paths, partial tiles) against torch."""
    from marl_amd import ops
    rng = np.random.RandomState(1234)
    g = torch.Generator().manual_seed(99)
    for case in range(40):
        M = int(rng.choice([1, 15, 16, 17, 100, 129, 300, 700]))
        N = int(rng.choice([1, 5, 16, 33, 64, 70, 130]))
        K0 = int(rng.choice([1, 3, 16, 20, 47, 64, 120, 130]))
        pad = int(rng.choice([0, 0, 1, 3]))                       # row stride K0 + pad: unaligned rows when odd
        off = int(rng.choice([0, 0, 1]))                          # base pointer offset in floats
        K1 = int(rng.choice([0, 0, 9]))
        NH, HW = (int(rng.choice([1, 3])), int(rng.choice([4, 11]))) if rng.rand() < 0.4 else (0, 0)
        NID = int(rng.choice([0, 0, 5]))
        base = torch.randn(M * (K0 + pad) + 4, generator=g)
        x0 = base[off:off + M * (K0 + pad)].view(M, K0 + pad)[:, :K0]
        parts = [x0]
        x1 = torch.randn(M, K1, generator=g) if K1 else None
        if K1:
            parts.append(x1)
        idx = None
        if NH:
            idx = torch.randint(-1, HW, (M, NH), generator=g)
            oh = torch.zeros(M, NH, HW)
            for jj in range(NH):
                v = idx[:, jj] >= 0
                oh[v, jj, idx[v, jj]] = 1
            parts.append(oh.reshape(M, -1))
        if NID:
            parts.append(torch.eye(NID)[torch.arange(M) % NID])
        X = torch.cat(parts, 1)
        K = X.shape[1]
        W, b = torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
        act = int(rng.rand() < 0.5)
        bd = cu(base, dev)
        x0d = bd[off:off + M * (K0 + pad)].view(M, K0 + pad)[:, :K0]
        src = ops.src(x0d, cu(x1, dev) if K1 else None, cu(idx, dev, torch.int32) if NH else None, NH, HW, NID)
        Y = torch.full((M, N), 3.0, device=dev)
        ops.linear(src, cu(W, dev), cu(b, dev), Y, M, N, K, act=act)
        ref = F.linear(X, W, b)
        ref = torch.relu(ref) if act else ref
        close(Y, ref, 3e-4, 3e-4, msg="linear case %d M%d N%d K%d" % (case, M, N, K))
        dY = torch.randn(M, N, generator=g)
        gate = torch.randn(M, N, generator=g) if rng.rand() < 0.5 else None
        Gm = dY * (gate > 0) if gate is not None else dY
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ops.linear_wgrad(cu(dY, dev), src, dW, db, M, N, K, Yact=cu(gate, dev) if gate is not None else None)
        sc = max(1.0, M ** 0.5)
        close(dW / sc, (Gm.t() @ X) / sc, 3e-4, 3e-4, msg="wgrad case %d" % case)
        close(db / sc, Gm.sum(0) / sc, 3e-4, 3e-4, msg="bgrad case %d" % case)
        dX = torch.empty(M, K, device=dev)                        # dX = (dY * gate) W through the k-major path
        ops.linear(ops.src(cu(dY, dev), gate=cu(gate, dev) if gate is not None else None), cu(W, dev), None, dX, M, K, N,
                   w_kmajor=True)
        close(dX, Gm @ W, 3e-4, 3e-4, msg="dX case %d" % case)


def test_linear_concat_and_groups(dev):
    from marl_amd import ops
    g = torch.Generator().manual_seed(5)
    M, K0, K1, NH, HW, NID = 75, 20, 9, 3, 4, 5
    x0, x1 = torch.randn(M, K0, generator=g), torch.randn(M, K1, generator=g)
    idx = torch.randint(-1, HW, (M, NH), generator=g)
    oh = torch.zeros(M, NH, HW)
    for j in range(NH):
        v = idx[:, j] >= 0
        oh[v, j, idx[v, j]] = 1
    eye = torch.eye(NID)[torch.arange(M) % NID]
    Xfull = torch.cat([x0, x1, oh.reshape(M, -1), eye], 1)
    K = Xfull.shape[1]
    W, b = torch.randn(13, K, generator=g), torch.randn(13, generator=g)
    Y = torch.empty(M, 13, device=dev)
    s = ops.src(cu(x0, dev), cu(x1, dev), cu(idx, dev, torch.int32), NH, HW, NID)
    ops.linear(s, cu(W, dev), cu(b, dev), Y, M, 13, K)
    close(Y, F.linear(Xfull, W, b), 2e-4)
    # weight gradient of the same virtual input
    dY = torch.randn(M, 13, generator=g)
    dW, db = torch.zeros(13, K, device=dev), torch.zeros(13, device=dev)
    ops.linear_wgrad(cu(dY, dev), s, dW, db, M, 13, K)
    close(dW, dY.t() @ Xfull, 3e-4)
    close(db, dY.sum(0), 3e-4)
    # grouped: 4 heads, shared input, strided weights inside one flat buffer, strided output columns
    G, N, Kg = 4, 6, 40
    X = torch.randn(M, Kg, generator=g)
    per = N * Kg + N
    flat = torch.randn(G * per, generator=g)
    fd = cu(flat, dev)
    Y = torch.zeros(M, G * N, device=dev)
    grp = ops.group(G, w=per, b=per, y=N)
    ops.linear(ops.src(cu(X, dev)), fd[:N * Kg].view(N, Kg), fd[N * Kg:per], Y, M, N, Kg, act=1, grp=grp)
    for k in range(G):
        Wk, bk = flat[k * per:k * per + N * Kg].view(N, Kg), flat[k * per + N * Kg:(k + 1) * per]
        close(Y[:, k * N:(k + 1) * N], torch.relu(F.linear(X, Wk, bk)), 2e-4, msg="group %d" % k)
    # grouped wgrad with relu gate
    dYg = torch.randn(M, G * N, generator=g)
    gflat = torch.zeros(G * per, device=dev)
    ops.linear_wgrad(cu(dYg, dev), ops.src(cu(X, dev)), gflat[:N * Kg].view(N, Kg), gflat[N * Kg:per], M, N, Kg,
                     Yact=Y, grp=ops.group(G, w=per, b=per, y=N, m0=N))
    Yc = Y.cpu()
    for k in range(G):
        Gk = dYg[:, k * N:(k + 1) * N] * (Yc[:, k * N:(k + 1) * N] > 0)
        close(gflat[k * per:k * per + N * Kg].view(N, Kg), Gk.t() @ X, 3e-4)
        close(gflat[k * per + N * Kg:(k + 1) * per], Gk.sum(0), 3e-4)


@pytest.mark.parametrize("M,N,K0,HW,gate", [(5000, 78, 64, 14, True), (3000, 78, 78, 0, False), (2500, 64, 64, 0, True),
                                            (4097, 80, 79, 0, False), (2100, 33, 20, 5, True), (9000, 1, 64, 0, False)])
def test_wgrad_full_width(dev, M, N, K0, HW, gate):
    """Narrow layers (N <= 80, K + 1 <= 80; QTRAN's 78-wide encoders, 64-wide heads) take the one-pass full-width
    weight-gradient kernel when dY rows are 16-byte aligned (row stride padded to 4 floats): dW, db vs torch."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(M + N + K0)
    K = K0 + HW
    ld = lambda w: (w + 3) // 4 * 4
    X0 = torch.randn(M, K0, generator=g)
    idx = torch.randint(-1, HW, (M, 1), generator=g) if HW else None
    Xfull = X0
    if HW:
        oh = torch.zeros(M, HW)
        v = idx[:, 0] >= 0
        oh[v, idx[v, 0]] = 1
        Xfull = torch.cat([X0, oh], 1)
    dY = torch.randn(M, N, generator=g)
    Ya = torch.randn(M, N, generator=g)
    pad = lambda t: cu(torch.cat([t, torch.zeros(t.shape[0], ld(t.shape[1]) - t.shape[1])], 1), dev)[:, :t.shape[1]]
    xs = ops.src(pad(X0), idx=cu(idx, dev, torch.int32) if HW else None, nhot=1 if HW else 0, hot_w=HW)
    dW, db = torch.full((N, K), 0.5, device=dev), torch.full((N,), 0.25, device=dev)
    ops.linear_wgrad(pad(dY), xs, dW, db, M, N, K, Yact=pad(Ya) if gate else None)
    Gm = dY * (Ya > 0) if gate else dY
    close(dW - 0.5, Gm.t() @ Xfull, 2e-3, 1e-3)
    close(db - 0.25, Gm.sum(0), 2e-3, 1e-3)


@pytest.mark.parametrize("M,K", [(76800, 64), (9001, 32), (4100, 30), (5000, 252), (4096, 4)])
def test_wgrad_one_output_column(dev, M, K):
    """N = 1 layers (last layer of QtranQBase.q / QtranV.v, mixer.py:384-388 / :410-414; hyper_b2[2], mixer.py:44-46):
    the weighted column sum kernel; gradients accumulate."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(M + K)
    X, dY = torch.randn(M, K, generator=g), torch.randn(M, 1, generator=g)
    ld = (K + 3) // 4 * 4
    Xd = cu(torch.cat([X, torch.full((M, ld - K), float("nan"))], 1), dev)[:, :K] if ld != K else cu(X, dev)
    dW, db = torch.full((1, K), 0.5, device=dev), torch.full((1,), 0.25, device=dev)
    for rep in range(2):
        ops.linear_wgrad(cu(dY, dev), ops.src(Xd), dW, db, M, 1, K)
    ref = (dY.double().t() @ X.double()).float()
    scale = max(1.0, float(ref.abs().max()))
    close((dW - 0.5) / scale, 2.0 * ref / scale, 1e-4, 1e-4)
    close(db - 0.25, 2.0 * dY.sum(0), 1e-3, 1e-4)


@pytest.mark.parametrize("rows,S,NH,HW,N3,G,remap,nl", [(333, 120, 0, 0, 1, 10, False, 3), (1000, 120, 5, 11, 5, 10, False, 3),
                                                        (70, 24, 2, 3, 2, 3, False, 3), (4100, 120, 5, 11, 5, 4, True, 3),
                                                        (129, 72, 0, 0, 16, 2, False, 3), (50, 36, 4, 9, 3, 1, False, 3),
                                                        (777, 120, 0, 0, 5, 2, False, 2), (4100, 120, 0, 0, 5, 2, True, 2),
                                                        (65, 28, 1, 4, 7, 3, False, 2),
                                                        # K1 > 192 (kept activations only) and state widths that are not multiples of 4:
                                                        # QPLEX on MMM2 (state 322, 10 x 18 actions), on 3s5z (216 + 8 x 14), odd small shapes
                                                        (700, 322, 10, 18, 10, 3, False, 3), (333, 322, 0, 0, 1, 4, False, 3),
                                                        (500, 322, 0, 0, 10, 2, False, 2), (129, 30, 2, 5, 3, 2, False, 3),
                                                        (1230, 216, 8, 14, 8, 2, True, 3), (90, 250, 0, 0, 4, 1, False, 3),
                                                        (77, 322, 10, 18, 10, 2, False, 3), (60, 203, 1, 7, 2, 2, False, 2),
                                                        # wide outputs (two-layer hypernet heads of QMIX with two_hyper_layers, mixer.py:36-43)
                                                        (777, 120, 0, 0, 160, 1, False, 2), (500, 120, 0, 0, 32, 1, False, 2),
                                                        (300, 322, 0, 0, 160, 2, False, 2), (260, 216, 0, 0, 128, 2, False, 2),
                                                        (100, 120, 0, 0, 20, 3, False, 2), (4100, 120, 0, 0, 160, 1, True, 2)])
def test_mlp3_fused(dev, rows, S, NH, HW, N3, G, remap, nl):
    """Fused three-layer heads (QPLEX lambda-net families, mixer.py:117-145) vs torch-CPU autograd: outputs and all six
    parameter gradients of every head; x = [state | one-hot actions] with ragged sizes, 'no action' indices and
    (remap) (T+1)-slot state storage read through an episode map."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(rows + S + NH + N3)
    K1 = S + NH * HW
    if remap:
        T, E = 41, 100
        B = rows // T
        rows = B * T
        store = torch.randn(E, T + 1, S, generator=g)
        emap = torch.randperm(E, generator=g)[:B]
        x0 = store[emap][:, 1:T + 1].reshape(rows, S)
        x0_src = ops.Rows(cu(store.reshape(-1, S), dev), (T, T + 1, 1), cu(emap, dev, torch.int32))
    else:
        x0 = torch.randn(rows, S, generator=g)
        # S = 36 / 72: rows at an odd stride - no 16-byte loads, every chunk takes the element path
        # S = 322 with rows = 700 / 333 / 500: rows padded to 16 bytes (how the replay stores MMM2 states) - 20 vector chunks
        if S % 8 == 0:
            x0_src = cu(x0, dev)
        elif S == 322 and rows > 100:
            x0_src = cu(torch.cat([x0, torch.full((rows, 2), float("nan"))], 1), dev)[:, :S]
        else:
            x0_src = cu(torch.cat([x0, x0[:, :1]], 1), dev)[:, :S]
    parts = [x0]
    idx = None
    if NH:
        idx = torch.randint(-1, HW, (rows, NH), generator=g)
        oh = torch.zeros(rows, NH, HW)
        for j in range(NH):
            v = idx[:, j] >= 0
            oh[v, j, idx[v, j]] = 1
        parts.append(oh.reshape(rows, -1))
    X = torch.cat(parts, 1)
    sizes = [(64, K1), (64,), (64, 64), (64,), (N3, 64), (N3,)] if nl == 3 else [(64, K1), (64,), (N3, 64), (N3,)]
    pad = lambda n: (n + 3) // 4 * 4
    per = sum(pad(int(np.prod(z))) for z in sizes)
    flat = torch.randn(G * per, generator=g) * 0.2
    fd, gd = cu(flat, dev), torch.zeros(G * per, device=dev)

    def views(buf, k):
        out, off = [], k * per
        for z in sizes:
            n = int(np.prod(z))
            out.append(buf[off:off + n].view(z))
            off += pad(n)
        return out

    class L:      # stands in for nn.Linear: .weight / .bias with .data and .grad
        def __init__(self, w, b, gw, gb):
            self.weight, self.bias = torch.nn.Parameter(w, requires_grad=False), torch.nn.Parameter(b, requires_grad=False)
            self.weight.grad, self.bias.grad = gw, gb
    heads = []
    for k in range(G):
        w, gr = views(fd, k), views(gd, k)
        heads.append([L(w[2 * i], w[2 * i + 1], gr[2 * i], gr[2 * i + 1]) for i in range(nl)])
    xs = ops.src(x0_src, idx=cu(idx, dev, torch.int32) if NH else None, nhot=NH, hot_w=HW)
    assert ops.mlp3_supported(xs, K1, 64, 64 if nl == 3 else 0, N3, G)
    Y = torch.full((rows, G * N3), 7.0, device=dev)
    kept_only = ops.mlp3_needs_kept(xs, K1, N3)      # K1 > 192 or wide outputs: no recomputing backward
    assert kept_only == ((K1 + (-S) % 4 + 15) // 16 > 12 or N3 > 16)
    hs = torch.full((ops.mlp3_save_floats(rows, nl == 3, G),), float("nan"), device=dev)
    ops.mlp3_fwd(ops.mlp3_weights(heads), xs, Y, rows, K1, N3, G, hsave=hs if kept_only else None)
    dY = torch.randn(rows, G * N3, generator=g)
    if kept_only:
        with pytest.raises(Exception):
            ops.mlp3_bwd(ops.mlp3_weights(heads), xs, cu(dY, dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G)
    for rep in range(2):      # gradients ACCUMULATE: the second call doubles them
        ops.mlp3_bwd(ops.mlp3_weights(heads), xs, cu(dY, dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G,
                     hsave=hs if kept_only else None)
    for k in range(G):
        ps = [v.clone().requires_grad_(True) for v in views(flat, k)]
        h = torch.relu(F.linear(X, ps[0], ps[1]))
        if nl == 3:
            h = torch.relu(F.linear(h, ps[2], ps[3]))
        y = F.linear(h, ps[-2], ps[-1])
        close(Y[:, k * N3:(k + 1) * N3], y, 2e-4, msg="head %d out" % k)
        y.backward(dY[:, k * N3:(k + 1) * N3])
        for name, pr, gv in zip(("W1", "b1", "W2", "b2", "W3", "b3") if nl == 3 else ("W1", "b1", "W3", "b3"), ps, views(gd, k)):
            scale = max(1.0, float(pr.grad.abs().max()))
            close(gv / scale, 2.0 * pr.grad / scale, 3e-4, 1e-4, msg="head %d d%s" % (k, name))
    assert not torch.isnan(gd).any()
    if kept_only:
        return
    # the pair that KEEPS h1 / h2 between forward and backward (marl_mlp3_fwd_save / marl_mlp3_bwd_saved): the kept values are the
    # values the backward would recompute - outputs bit-identical; the gradients are the same per-stripe sums added over another
    # number of stripes (the kept backward runs two workgroups per CU)
    Y2, gd_rec = torch.full((rows, G * N3), 7.0, device=dev), gd.clone()
    ops.mlp3_fwd(ops.mlp3_weights(heads), xs, Y2, rows, K1, N3, G, hsave=hs)
    assert torch.equal(Y2, Y)
    gd.zero_()
    for rep in range(2):
        ops.mlp3_bwd(ops.mlp3_weights(heads), xs, cu(dY, dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G, hsave=hs)
    scale = float(gd_rec.abs().max())
    close(gd / scale, gd_rec / scale, 2e-6, 1e-5)
    assert not torch.isnan(gd).any()
    gd2 = gd.clone()
    gd.zero_()
    for rep in range(2):      # run to run: bitwise
        ops.mlp3_bwd(ops.mlp3_weights(heads), xs, cu(dY, dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G, hsave=hs)
    assert torch.equal(gd, gd2)


@pytest.mark.parametrize("rows,S,NH,HW,N3,G,remap", [(333, 120, 0, 0, 1, 10, False), (1000, 120, 5, 11, 5, 10, False),
                                                     (70, 24, 2, 3, 2, 3, False), (4100, 120, 5, 11, 5, 4, True),
                                                     (129, 72, 0, 0, 16, 2, False), (50, 36, 4, 9, 3, 1, False),
                                                     (2100, 120, 0, 0, 5, 2, True), (129, 30, 2, 5, 3, 2, False),
                                                     (40000, 120, 5, 11, 5, 10, False)])
def test_mlp3_x6_split(dev, rows, S, NH, HW, N3, G, remap):
    """bf16x6 split pair of the fused three-layer heads (csrc/mlp3_x6.hip, opt-in gemm_mode) vs an fp64 evaluation on the CPU:
    outputs and all six parameter gradients of every head, at the SAME bounds test_mlp3_fused uses for the fp32 MFMA pair -
    and the errors against fp64 side by side with the fp32 pair's: measured 1.0-1.6 x the fp32 MFMA path's up to 4 000 rows and
    2-3 x at 40 000 rows (3.6e-6 vs 1.3e-6 of the tensor's scale, worst on the bias gradients, which ride on the weight-gradient
    products here and are plain fp32 sums there); asserted: not worse than 4 x, or 5e-6 of the scale."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(rows + S + NH + N3)
    K1 = S + NH * HW
    if remap:
        T, E = 41, 120
        B = rows // T
        rows = B * T
        store = torch.randn(E, T + 1, S, generator=g)
        emap = torch.randperm(E, generator=g)[:B]
        x0 = store[emap][:, 1:T + 1].reshape(rows, S)
        x0_src = ops.Rows(cu(store.reshape(-1, S), dev), (T, T + 1, 1), cu(emap, dev, torch.int32))
    else:
        x0 = torch.randn(rows, S, generator=g)
        x0_src = cu(x0, dev) if S % 8 == 0 else cu(torch.cat([x0, x0[:, :1]], 1), dev)[:, :S]
    parts = [x0]
    idx = None
    if NH:
        idx = torch.randint(-1, HW, (rows, NH), generator=g)
        oh = torch.zeros(rows, NH, HW)
        for j in range(NH):
            v = idx[:, j] >= 0
            oh[v, j, idx[v, j]] = 1
        parts.append(oh.reshape(rows, -1))
    X = torch.cat(parts, 1)
    sizes = [(64, K1), (64,), (64, 64), (64,), (N3, 64), (N3,)]
    pad = lambda n: (n + 3) // 4 * 4
    per = sum(pad(int(np.prod(z))) for z in sizes)
    flat = torch.randn(G * per, generator=g) * 0.2
    fd, gd = cu(flat, dev), torch.zeros(G * per, device=dev)

    def views(buf, k):
        out, off = [], k * per
        for z in sizes:
            n = int(np.prod(z))
            out.append(buf[off:off + n].view(z))
            off += pad(n)
        return out

    class L:
        def __init__(self, w, b, gw, gb):
            self.weight, self.bias = torch.nn.Parameter(w, requires_grad=False), torch.nn.Parameter(b, requires_grad=False)
            self.weight.grad, self.bias.grad = gw, gb
    heads = []
    for k in range(G):
        w, gr = views(fd, k), views(gd, k)
        heads.append([L(w[2 * i], w[2 * i + 1], gr[2 * i], gr[2 * i + 1]) for i in range(3)])
    xs = ops.src(x0_src, idx=cu(idx, dev, torch.int32) if NH else None, nhot=NH, hot_w=HW)
    assert ops.mlp3_x6_supported(xs, K1, 64, 64, N3, G)
    dY = torch.randn(rows, G * N3, generator=g)
    # relu kinks: a pre-activation within rounding of zero may gate differently in two correct evaluations - such rows (a handful
    # among 10^7 pre-activations of the large case) get a zero output gradient for that head on both sides
    for k in range(G):
        ps = [v.double() for v in views(flat, k)]
        p1 = F.linear(X.double(), ps[0], ps[1])
        p2 = F.linear(torch.relu(p1), ps[2], ps[3])
        kink = ((p1.abs() < 2e-6).any(1) | (p2.abs() < 2e-6).any(1))
        dY[kink, k * N3:(k + 1) * N3] = 0
    res = {}
    for x6 in (True, False):
        Y = torch.full((rows, G * N3), 7.0, device=dev)
        hs = torch.full((ops.mlp3_save_floats(rows, True, G),), float("nan"), device=dev)
        ops.mlp3_fwd(ops.mlp3_weights(heads), xs, Y, rows, K1, N3, G, hsave=hs, x6=x6)
        if x6:      # without keeping (target mixer): the same outputs
            Y0 = torch.full((rows, G * N3), 7.0, device=dev)
            ops.mlp3_fwd(ops.mlp3_weights(heads), xs, Y0, rows, K1, N3, G, x6=True)
            assert torch.equal(Y0, Y)
        gd.zero_()
        for rep in range(2):      # gradients ACCUMULATE: the second call doubles them
            ops.mlp3_bwd(ops.mlp3_weights(heads), xs, cu(dY, dev), ops.mlp3_weights(heads, grad=True), rows, K1, N3, G, hsave=hs, x6=x6)
        assert not torch.isnan(gd).any()
        res[x6] = (Y.cpu().double(), gd.cpu().double().clone())
    worst = {True: 0.0, False: 0.0}
    for k in range(G):
        ps = [v.double().clone().requires_grad_(True) for v in views(flat, k)]
        h = torch.relu(F.linear(X.double(), ps[0], ps[1]))
        h = torch.relu(F.linear(h, ps[2], ps[3]))
        y = F.linear(h, ps[4], ps[5])
        y.backward(dY[:, k * N3:(k + 1) * N3].double())
        for x6 in (True, False):
            Yr, gr = res[x6]
            err = float((Yr[:, k * N3:(k + 1) * N3] - y.detach()).abs().max() / y.detach().abs().max())
            if x6:
                close(Yr[:, k * N3:(k + 1) * N3].float(), y.detach().float(), 2e-4, msg="head %d out" % k)
            worst[x6] = max(worst[x6], err)
            for name, pr, gv in zip(("W1", "b1", "W2", "b2", "W3", "b3"), ps, views(gr, k)):
                scale = max(1.0, float(pr.grad.abs().max()))
                if x6:
                    close((gv / scale).float(), (2.0 * pr.grad / scale).float(), 3e-4, 1e-4, msg="head %d d%s" % (k, name))
                worst[x6] = max(worst[x6], float((gv - 2.0 * pr.grad).abs().max() / max(float(pr.grad.abs().max()), 1e-30) / 2.0))
    # against fp64 the split is as good as the fp32 MFMA path (relative to each tensor's scale)
    print("mlp3 rows=%d K1=%d: worst error / scale vs fp64: bf16x6 %.2e, fp32 MFMA %.2e" % (rows, K1, worst[True], worst[False]))
    assert worst[True] <= max(4.0 * worst[False], 5e-6)


@pytest.mark.parametrize("B,O", [(6, 24), (30, 24), (20, 80), (17, 116), (21, 64), (19, 52), (18, 128), (17, 176), (17, 148)])
def test_wgrad_large_rows_and_remap(dev, B, O):
    """B >= 17 (M >= 4096 rows, 64 outputs) takes the direct no-LDS kernel (O >= 128: in column passes), B = 6 the
    LDS-staged one."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(9 + B + O)
    T, N = 50, 5                         # (T+1)-slot storage read through the row remap
    store = torch.randn(B, T + 1, N, O, generator=g)
    u = torch.randint(-1, 7, (B, T, N), generator=g)
    dY = torch.randn(B * T * N, 64, generator=g)
    for t0, uoff in ((0, -1), (1, 0)):
        xs = store[:, t0:t0 + T].reshape(B * T * N, O)
        up = torch.full((B, T, N), -1, dtype=torch.long)
        if uoff == -1:
            up[:, 1:] = u[:, :-1]
        else:
            up = u.clone()
        oh = torch.zeros(B * T * N, 7)
        flat = up.reshape(-1)
        oh[flat >= 0, flat[flat >= 0]] = 1
        Xfull = torch.cat([xs, oh, torch.eye(N)[torch.arange(B * T * N) % N]], 1)
        K = Xfull.shape[1]
        s = ops.src(cu(store.reshape(-1, O), dev), idx=cu(u.reshape(-1, 1), dev, torch.int32), nhot=1, hot_w=7, nid=N,
                    remap0=(T * N, (T + 1) * N, t0 * N), remapi=(T * N, T * N, uoff * N))
        dW, db = torch.zeros(64, K, device=dev), torch.zeros(64, device=dev)
        ops.linear_wgrad(cu(dY, dev), s, dW, db, B * T * N, 64, K)
        close(dW, dY.t() @ Xfull, 2e-3, 1e-3)
        close(db, dY.sum(0), 2e-3, 1e-3)
        Y = torch.empty(B * T * N, 10, device=dev)
        W = torch.randn(10, K, generator=g)
        ops.linear(s, cu(W, dev), None, Y, B * T * N, 10, K)
        close(Y, Xfull @ W.t(), 3e-4)


@pytest.mark.parametrize("M,S,E_", [(5000, 216, 78), (4100, 200, 64), (9000, 224, 20), (4500, 120, 78)])
def test_wgrad_split_state_and_encoder_columns(dev, M, S, E_):
    """First layer of QTRAN's joint-Q / V heads (network/mixer.py:378-388): dW = dy1^T [s | esum] as two reductions into column
    blocks of ONE weight-gradient tensor - the state block (up to 224 dense columns: the wide instantiation of the LDS-staged
    tall kernel) with the bias gradient, the encoder block (a width that is not a multiple of 4) without; gradients accumulate."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(M + S)
    s, e = torch.randn(M, S, generator=g), torch.randn(M, (E_ + 3) // 4 * 4, generator=g)[:, :E_]
    dY = torch.randn(M, 64, generator=g)
    base_w, base_b = torch.randn(64, S + E_, generator=g), torch.randn(64, generator=g)
    gw, gb = cu(base_w, dev), cu(base_b, dev)
    ed = cu(torch.cat([e, torch.zeros(M, (-E_) % 4)], 1), dev)[:, :E_]       # rows padded to 16 bytes, as the kernels' scratch is
    ops.linear_wgrad(cu(dY, dev), ops.src(cu(s, dev)), gw[:, :S], gb, M, 64, S)
    ops.linear_wgrad(cu(dY, dev), ops.src(ed), gw[:, S:], None, M, 64, E_)
    ref = dY.double().t() @ torch.cat([s, e], 1).double()
    sc = float(ref.abs().max())
    close((gw.cpu() - base_w) / sc, ref.float() / sc, 1e-4, 1e-4, msg="dW")
    close(gb.cpu() - base_b, dY.sum(0), 1e-4 * float(dY.sum(0).abs().max()), 1e-4, msg="db")


# ------------------------------------------------------------------------------------- agent
def _agent_case(shape, B, T, dev, seed=0, with_h0=False):
    args = seeded.make_args(shape, "qmix", episode_limit=T)
    p_np = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11 + seed, scale=2.0)
    rng = np.random.default_rng(seed)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    obs = rng.standard_normal((B, T, N, O)).astype(np.float32)
    ufed = rng.integers(-1, A, size=(B, T, N))
    h0 = rng.standard_normal((B * N, 64)).astype(np.float32) * 0.5 if with_h0 else None
    return args, p_np, obs, ufed, h0


def _oracle_unroll(args, p_np, obs, ufed, h0, requires_grad=False):
    p = {k: torch.tensor(v, requires_grad=requires_grad) for k, v in p_np.items()}
    B, T, N, O = obs.shape
    A = args.n_actions
    oh = np.zeros((B, T, N, A), np.float32)
    bb, tt, nn = np.nonzero(ufed >= 0)
    oh[bb, tt, nn, ufed[bb, tt, nn]] = 1
    h = torch.zeros(B * N, 64) if h0 is None else torch.tensor(h0)
    q, hs, hl = nets.agent_unroll(p, torch.tensor(obs), torch.tensor(oh), h)
    return p, q, hs, hl


@pytest.mark.parametrize("shape,B,T,with_h0", [("2s3z", 7, 5, False), ("2s3z", 70, 3, True), ("matrix", 9, 1, False),
                                               ("3s5z", 5, 4, True), ("MMM2", 4, 3, False)])
def test_agent_unroll_fwd(dev, shape, B, T, with_h0):
    from marl_amd import ops
    args, p_np, obs, ufed, h0 = _agent_case(shape, B, T, dev, with_h0=with_h0)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    with torch.no_grad():
        _, q_ref, hs_ref, hl_ref = _oracle_unroll(args, p_np, obs, ufed, h0)
    pd = {k: cu(v, dev) for k, v in p_np.items()}
    w = ops.agent_weights(pd)
    q = torch.empty(B, T, N, A, device=dev)
    hs = torch.empty(B, T, N, 64, device=dev)
    hl = torch.empty(B * N, 64, device=dev)
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)  # opaque tile layout, 6 planes per row-step
    ops.agent_unroll_fwd(w, cu(obs, dev), T * N, 0, cu(ufed, dev, torch.int32), T * N, 0,
                         cu(h0, dev) if h0 is not None else None, q, hs, hl, saved, B, T, N, O, A)
    close(q, q_ref, 1e-4, msg="q")
    close(hs, hs_ref, 1e-4, msg="hs")
    close(hl, hl_ref, 1e-4, msg="h_last")
    # saved plane 0 is the hidden state fed INTO each step
    hprev = torch.cat([(torch.zeros(B, 1, N, 64) if h0 is None else torch.tensor(h0).view(B, 1, N, 64)), hs_ref[:, :-1]], 1)
    # (and one more slab: the hidden state after the last step, where the backward pass looks for h(T-1))
    hprev = torch.cat([hprev, hs_ref[:, -1:]], 1)
    close(ops.saved_plane(saved, 0, B * N).reshape(T + 1, B, N, 64).permute(1, 0, 2, 3), hprev, 1e-4, msg="hprev")


def test_agent_unroll_shifted_storage(dev):
    """(T+1)-slot observation storage + one-step shifted last action == the o / o_next passes."""
    from marl_amd import ops
    B, T = 5, 4
    args, p_np, _, _, _ = _agent_case("2s3z", B, T, dev)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    rng = np.random.default_rng(3)
    store = rng.standard_normal((B, T + 1, N, O)).astype(np.float32)
    u = rng.integers(-1, A, size=(B, T, N))
    pd = {k: cu(v, dev) for k, v in p_np.items()}
    w = ops.agent_weights(pd)
    sd, ud = cu(store, dev), cu(u, dev, torch.int32)
    for t0, ut0 in ((0, -1), (1, 0)):
        ufed = np.full((B, T, N), -1)
        if ut0 == -1:
            ufed[:, 1:] = u[:, :-1]
        else:
            ufed = u
        with torch.no_grad():
            _, q_ref, hs_ref, _ = _oracle_unroll(args, p_np, store[:, t0:t0 + T], ufed, None)
        q = torch.empty(B, T, N, A, device=dev)
        ops.agent_unroll_fwd(w, sd, (T + 1) * N, t0, ud, T * N, ut0, None, q, None, None, None, B, T, N, O, A)
        close(q, q_ref, 1e-4)


@pytest.mark.parametrize("B,T,with_h0,ragged,ut0", [(1700, 6, True, True, 0), (2100, 5, False, True, -1), (4096, 4, True, False, 0),
                                                     (4100, 4, True, True, 0), (9000, 4, False, True, 0), (2621, 7, True, True, -1)])
def test_agent_unroll_x6_plain_round6_decomposition(dev, B, T, with_h0, ragged, ut0):
    """csrc/agent_x6p.hip (non-saving split unrolls of more than 512 row tiles: the recurrent team runs x W_ih + h W_hh down one
    accumulator chain, three to five row tiles per workgroup, two barriers per step) against the CPU oracle at the unroll bound (1e-4)
    and beside csrc/agent_x6.hip on the same inputs (unroll_r6 = 0; the two differ only in where the last action's fc1 column is added):
    (T+1)-slot storage read through an episode map, the shifted / unshifted fed action, ragged episode lengths, a carried hidden state,
    three / four / five tiles per workgroup, several rounds of workgroups, a partial last workgroup."""
    from marl_amd import ops, experiments
    args, p_np, _, _, _ = _agent_case("2s3z", B, T, dev, with_h0=with_h0)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    assert ops.agent_unroll_x6_plain_r6(B, T, N, O, A) and not ops.agent_unroll_x6_plain_r6(B, T, N, O, A, cu_budget=128)
    assert not ops.agent_unroll_x6_plain_r6(1600, T, N, O, A)            # 500 row tiles: one round of two-tile workgroups of agent_x6.hip
    rng = np.random.default_rng(B + T)
    E = B + 2
    store = rng.standard_normal((E, T + 1, N, O)).astype(np.float32)
    u = rng.integers(-1, A, size=(B, T, N))
    emap = rng.permutation(E)[:B]
    lens = rng.integers(1, T + 1, size=B) if ragged else np.full(B, T)
    lens[0] = T
    h0 = (rng.standard_normal((B * N, 64)).astype(np.float32) * 0.3) if with_h0 else None
    w = ops.agent_weights({k: cu(v, dev) for k, v in p_np.items()})
    sd, ud, ed, ld = cu(store, dev), cu(u, dev, torch.int32), cu(emap, dev, torch.int32), cu(lens, dev, torch.int32)
    t0 = 1 if ut0 == 0 else 0
    outs = {}
    for r6 in (1, 0):
        with experiments.override(unroll_r6=r6):
            assert ops.agent_unroll_x6_plain_r6(B, T, N, O, A) == bool(r6)
            q, hl = torch.full((B, T, N, A), 9.0, device=dev), torch.full((B * N, 64), 9.0, device=dev)
            ops.agent_unroll_fwd_x6(w, sd, (T + 1) * N, t0, ud, T * N, ut0, cu(h0, dev) if h0 is not None else None, q, None, hl, None,
                                    B, T, N, O, A, ep_len=ld, ep_map=ed)
            outs[r6] = (q.cpu(), hl.cpu())
    obs = store[emap][:, t0:t0 + T].copy()
    for b in range(B):
        obs[b, lens[b]:] = 0          # steps t >= ep_len feed zeros
    ufed = u.copy()
    if ut0 == -1:
        ufed[:, 1:] = u[:, :-1]; ufed[:, 0] = -1
    with torch.no_grad():
        _, q_ref, _, hl_ref = _oracle_unroll(args, p_np, obs, ufed, h0)
    for r6 in (1, 0):
        close(outs[r6][0], q_ref, 1e-4, msg="q (unroll_r6 = %d)" % r6)
        close(outs[r6][1], hl_ref, 1e-4, msg="h_last (unroll_r6 = %d)" % r6)
    close(outs[1][0], outs[0][0], 2e-5, msg="the two split unrolls")
    e6 = float((outs[1][0].double() - q_ref.double()).abs().max())
    e5 = float((outs[0][0].double() - q_ref.double()).abs().max())
    print("plain unroll B=%d T=%d: max |q - oracle|: agent_x6p %.2e, agent_x6 %.2e" % (B, T, e6, e5))


@pytest.mark.parametrize("shape,B,T,cus", [("2s3z", 37, 5, 4), ("2s3z", 700, 6, 48), ("3s5z", 40, 4, 8), ("2s3z", 9, 2, 2),
                                            ("2s3z", 37, 5, 16), ("2s3z", 300, 7, 256), ("3s5z", 21, 4, 64),
                                            ("MMM2", 60, 4, 16), ("MMM2", 1000, 3, 256), ("MMM2", 30, 5, 128)])
def test_double_q_unroll_reuses_input_side_work_bitwise(dev, shape, B, T, cus):
    """gi_out / gi_in (include/marl_hip.h): an unroll over steps 1..T of (T+1)-slot storage that READS the input-side gate sums
    an unroll over steps 0..T-1 stored == the same unroll computing everything, bit for bit - with ragged episode lengths
    (steps ep_len - 1 and T - 1 are computed in full), an episode map, a carried hidden state and a partial last row tile.
    A small CU budget gives several row tiles per workgroup at test sizes (the multi-tile kernel); the last three 2s3z / 3s5z
    cases run one tile per workgroup (the software-pipelined kernel).  MMM2 (176-wide observations, 18 actions): three row
    tiles per workgroup are more rows than the prefetch registers of the reading launch cover (it fetches the rest when it
    refills the tile), and the launch it is compared with is the half-tile prefetch kernel."""
    from marl_amd import ops
    args, p_np, _, _, _ = _agent_case(shape, B, T, dev)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    assert ops.agent_unroll_reuse_supported(B, T, N, O, A, cus)
    rng = np.random.default_rng(B + T)
    E = B + 3                                                  # storage holds more episodes than the batch
    store = cu(rng.standard_normal((E, T + 1, N, O)).astype(np.float32), dev)
    u = cu(rng.integers(-1, A, size=(B, T, N)), dev, torch.int32)
    emap = cu(rng.permutation(E)[:B], dev, torch.int32)
    lens = rng.integers(1, T + 1, size=B)
    lens[0], lens[-1] = T, 1
    ep_len = cu(lens, dev, torch.int32)
    w = ops.agent_weights({k: cu(v, dev) for k, v in p_np.items()})
    H = 64
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
    gi = torch.full(ops.saved_shape(T, B, N, planes=3), float("nan"), device=dev)
    q0, h_last = torch.empty(B, T, N, A, device=dev), torch.empty(B * N, H, device=dev)
    # eval pass: slots 0..T-1, last action = the previous step's (u_t0 = -1), stores activations and gate sums
    ops.agent_unroll_fwd(w, store, (T + 1) * N, 0, u, T * N, -1, None, q0, None, h_last, saved, B, T, N, O, A,
                         ep_len=ep_len, ep_map=emap, cu_budget=cus, gi_out=gi)
    outs = []
    for reuse in (True, False):
        q, hl = torch.empty(B, T, N, A, device=dev), torch.empty(B * N, H, device=dev)
        ops.agent_unroll_fwd(w, store, (T + 1) * N, 1, u, T * N, 0, h_last, q, None, hl, None, B, T, N, O, A,
                             ep_len=ep_len, ep_map=emap, cu_budget=cus, gi_in=gi if reuse else None)
        outs.append((q.cpu(), hl.cpu()))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("shape,B,T,cus", [("2s3z", 37, 5, 4), ("2s3z", 300, 7, 16), ("3s5z", 40, 4, 8), ("MMM2", 60, 4, 16),
                                            ("MMM2", 1024, 3, 256), ("MMM2", 30, 5, 128)])
def test_saving_unroll_variants_equal_register_prefetch_bitwise(dev, shape, B, T, cus):
    """Variants of the activation-saving unroll == the plain four-register prefetch kernel, bit for bit (q, the final hidden
    state, all six saved planes and the input-side gate sums; ragged episode lengths - rows past their end feed zeros -, an
    episode map, (T+1)-slot storage with the shifted last action, a partial last row tile):
      * experiments fwd_dma = 1 (MARL_FWD_DMA=1): observation tile and fed-back actions filled by LDS-DMA (opt-in experiment);
      * the default for wide observations with two action tiles (MMM2 at >= 3 row tiles per workgroup: six prefetch
        registers, fc2 fragments in LDS) against fwd_w2l = 0."""
    import os
    from marl_amd import ops
    args, p_np, _, _, _ = _agent_case(shape, B, T, dev)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    rng = np.random.default_rng(B + T)
    E = B + 2
    store = cu(rng.standard_normal((E, T + 1, N, O)).astype(np.float32), dev)
    u = cu(rng.integers(-1, A, size=(B, T, N)), dev, torch.int32)
    emap = cu(rng.permutation(E)[:B], dev, torch.int32)
    lens = rng.integers(1, T + 1, size=B)
    lens[0], lens[-1] = T, 1
    ep_len = cu(lens, dev, torch.int32)
    h0 = cu(rng.standard_normal((B * N, 64)).astype(np.float32) * 0.3, dev)
    w = ops.agent_weights({k: cu(v, dev) for k, v in p_np.items()})
    outs = {}
    from marl_amd import experiments
    for mode in ("0", "1", "w2l"):
        with experiments.override(fwd_dma=0 if mode == "w2l" else int(mode), fwd_w2l=1 if mode == "w2l" else 0):
            for t0, ut0 in ((0, -1), (1, 0)):
                saved = torch.zeros(ops.saved_shape(T, B, N), device=dev)
                gi = torch.zeros(ops.saved_shape(T, B, N, planes=3), device=dev)
                q, hl = torch.empty(B, T, N, A, device=dev), torch.empty(B * N, 64, device=dev)
                ops.agent_unroll_fwd(w, store, (T + 1) * N, t0, u, T * N, ut0, h0, q, None, hl, saved, B, T, N, O, A,
                                     ep_len=ep_len, ep_map=emap, cu_budget=cus, gi_out=gi)
                rows = B * N
                planes = [ops.saved_plane(saved, k, rows).cpu() for k in range(6)] + [ops.saved_plane(gi, k, rows).cpu() for k in range(3)]
                outs[(mode, t0)] = (q.cpu(), hl.cpu(), planes)
    for t0 in (0, 1):
        for other in ("1", "w2l"):
            a, b = outs[("0", t0)], outs[(other, t0)]
            assert torch.isfinite(a[0]).all()
            assert torch.equal(a[0], b[0]), "%s: q (t0=%d)" % (other, t0)
            assert torch.equal(a[1], b[1]), "%s: h_last" % other
            for k, (x, y) in enumerate(zip(a[2], b[2])):
                assert torch.equal(x, y), "%s: plane %d (t0=%d)" % (other, k, t0)


@pytest.mark.parametrize("shape,B,T,with_h0,ragged", [("2s3z", 7, 5, False, False), ("2s3z", 70, 6, True, True), ("2s3z", 3, 4, True, False),
                                                       ("2s3z", 900, 9, False, True), ("2s3z", 2100, 12, True, True),
                                                       ("3s5z", 9, 5, True, True), ("3s5z", 600, 4, False, True),
                                                       ("MMM2", 5, 5, True, True), ("MMM2", 300, 4, False, True)])
def test_agent_unroll_x6_split(dev, shape, B, T, with_h0, ragged):
    """Unroll that saves nothing on the bf16x6 split kernels (csrc/agent_x6.hip, opt-in gemm_mode): q, hs and the final hidden state
    against the CPU oracle at the bound of test_agent_unroll_fwd (1e-4), and beside the fp32 MFMA kernel on the same inputs -
    (T+1)-slot storage read through an episode map with the shifted last action, ragged episode lengths (rows past their end
    feed zeros), a carried hidden state, partial last row tile, one and two row tiles per workgroup (> 256 tiles)."""
    from marl_amd import ops
    # (3s5z: 150 input columns = five fc1 chunks; MMM2: 204 = seven, and 18 actions = two action tiles; one row tile per workgroup)
    args, p_np, _, _, _ = _agent_case(shape, B, T, dev, with_h0=with_h0)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    assert ops.agent_unroll_x6_supported(B, T, N, O, A)
    rng = np.random.default_rng(B + T)
    E = B + 2
    store = rng.standard_normal((E, T + 1, N, O)).astype(np.float32)
    u = rng.integers(-1, A, size=(B, T, N))
    emap = rng.permutation(E)[:B]
    lens = rng.integers(1, T + 1, size=B) if ragged else np.full(B, T)
    lens[0] = T
    h0 = (rng.standard_normal((B * N, 64)).astype(np.float32) * 0.3) if with_h0 else None
    w = ops.agent_weights({k: cu(v, dev) for k, v in p_np.items()})
    sd, ud, ed = cu(store, dev), cu(u, dev, torch.int32), cu(emap, dev, torch.int32)
    ld = cu(lens, dev, torch.int32)
    outs = {}
    for mode in ("x6", "f32"):
        q, hs, hl = torch.full((B, T, N, A), 9.0, device=dev), torch.full((B, T, N, 64), 9.0, device=dev), torch.full((B * N, 64), 9.0, device=dev)
        h0d = cu(h0, dev) if h0 is not None else None
        if mode == "x6":
            ops.agent_unroll_fwd_x6(w, sd, (T + 1) * N, 1, ud, T * N, 0, h0d, q, hs, hl, None, B, T, N, O, A, ep_len=ld, ep_map=ed)
        else:
            ops.agent_unroll_fwd(w, sd, (T + 1) * N, 1, ud, T * N, 0, h0d, q, hs, hl, None, B, T, N, O, A, ep_len=ld, ep_map=ed)
        outs[mode] = (q.cpu(), hs.cpu(), hl.cpu())
    obs = store[emap][:, 1:T + 1].copy()
    for b in range(B):
        obs[b, lens[b]:] = 0          # steps t >= ep_len feed zeros
    with torch.no_grad():
        _, q_ref, hs_ref, hl_ref = _oracle_unroll(args, p_np, obs, u, h0)
    for mode in ("x6", "f32"):
        close(outs[mode][0], q_ref, 1e-4, msg=mode + " q")
        close(outs[mode][1], hs_ref, 1e-4, msg=mode + " hs")
        close(outs[mode][2], hl_ref, 1e-4, msg=mode + " h_last")
    e6 = float((outs["x6"][0].double() - q_ref.double()).abs().max())
    e32 = float((outs["f32"][0].double() - q_ref.double()).abs().max())
    print("agent unroll B=%d T=%d: max |q - oracle|: bf16x6 %.2e, fp32 MFMA %.2e" % (B, T, e6, e32))


@pytest.mark.parametrize("shape,B,T,cus", [("2s3z", 37, 5, 0), ("2s3z", 700, 6, 48), ("2s3z", 9, 4, 2), ("2s3z", 300, 7, 256), ("2s3z", 1700, 5, 128),
                                            ("3s5z", 21, 5, 0), ("3s5z", 400, 4, 128), ("MMM2", 13, 5, 0), ("MMM2", 200, 4, 128)])
def test_agent_unroll_x6_saved_planes_and_gate_sum_reuse(dev, shape, B, T, cus):
    """The activation-saving and the gate-sum-reading variants of the split unroll (csrc/agent_x6.hip):
      * the six saved planes, the stored input-side sums, q and the final hidden state of the eval pass == the fp32 MFMA
        kernel's within 1e-4 (same tile layout: decoded with ops.saved_plane);
      * the continuation over slots 1..T that READS the stored sums == the same launch computing everything, bit for bit
        (ragged episode lengths: steps ep_len - 1 and T - 1 are computed in full; episode map; carried hidden state;
        partial last row tile; one and two row tiles per workgroup by CU budget)."""
    from marl_amd import ops
    args, p_np, _, _, _ = _agent_case(shape, B, T, dev)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    assert ops.agent_unroll_x6_supported(B, T, N, O, A)
    rng = np.random.default_rng(B + T)
    E = B + 3
    store = cu(rng.standard_normal((E, T + 1, N, O)).astype(np.float32), dev)
    u = cu(rng.integers(-1, A, size=(B, T, N)), dev, torch.int32)
    emap = cu(rng.permutation(E)[:B], dev, torch.int32)
    lens = rng.integers(1, T + 1, size=B)
    lens[0], lens[-1] = T, 1
    ep_len = cu(lens, dev, torch.int32)
    w = ops.agent_weights({k: cu(v, dev) for k, v in p_np.items()})
    rows = B * N
    ev = {}
    for mode in ("x6", "f32"):
        saved = torch.full(ops.saved_shape(T, B, N), float("nan"), device=dev)
        gi = torch.full(ops.saved_shape(T, B, N, planes=3), float("nan"), device=dev)
        q0, h_last = torch.empty(B, T, N, A, device=dev), torch.empty(B * N, 64, device=dev)
        fn = ops.agent_unroll_fwd_x6 if mode == "x6" else ops.agent_unroll_fwd
        fn(w, store, (T + 1) * N, 0, u, T * N, -1, None, q0, None, h_last, saved, B, T, N, O, A,
           ep_len=ep_len, ep_map=emap, cu_budget=cus, gi_out=gi)
        ev[mode] = (q0, h_last, saved, gi)
    close(ev["x6"][0], ev["f32"][0], 1e-4, msg="q")
    close(ev["x6"][1], ev["f32"][1], 1e-4, msg="h_last")
    for k in range(6):
        close(ops.saved_plane(ev["x6"][2], k, rows), ops.saved_plane(ev["f32"][2], k, rows), 1e-4, msg="saved plane %d" % k)
    if A <= 16:      # (with two action tiles the fp32 kernels store PRE-SCALED sums - gru_prescale - for their own readers; the split kernels plain ones)
        for k in range(3):
            close(ops.saved_plane(ev["x6"][3], k, rows), ops.saved_plane(ev["f32"][3], k, rows), 1e-4, msg="gate sums %d" % k)
    _, h_last, _, gi = ev["x6"]
    outs = []
    for reuse in (True, False):
        q, hl = torch.empty(B, T, N, A, device=dev), torch.empty(B * N, 64, device=dev)
        ops.agent_unroll_fwd_x6(w, store, (T + 1) * N, 1, u, T * N, 0, h_last, q, None, hl, None, B, T, N, O, A,
                                ep_len=ep_len, ep_map=emap, cu_budget=cus, gi_in=gi if reuse else None)
        outs.append((q.cpu(), hl.cpu()))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,T", [(7, 5), (40, 6), (700, 4)])
def test_agent_unroll_bwd_from_x6_saved(dev, B, T):
    """BPTT (the fp32 kernels) from the activations the split unroll saved: gradients vs torch autograd of the oracle unroll at
    the bounds of test_agent_unroll_bwd."""
    from marl_amd import ops
    args, p_np, obs, ufed, h0 = _agent_case("2s3z", B, T, dev, seed=1)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    I = O + A + N
    p, q_ref, hs_ref, _ = _oracle_unroll(args, p_np, obs, ufed, None, requires_grad=True)
    g = torch.Generator().manual_seed(2)
    dq = torch.randn(B, T, N, A, generator=g)
    dhs = torch.randn(B, T, N, 64, generator=g) * 0.3
    ((q_ref * dq).sum() + (hs_ref * dhs).sum()).backward()
    pd = {k: cu(v, dev) for k, v in p_np.items()}
    w = ops.agent_weights(pd)
    q, hs = torch.empty(B, T, N, A, device=dev), torch.empty(B, T, N, 64, device=dev)
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
    obs_d, u_d = cu(obs, dev), cu(ufed, dev, torch.int32)
    ops.agent_unroll_fwd_x6(w, obs_d, T * N, 0, u_d, T * N, 0, None, q, hs, None, saved, B, T, N, O, A)
    close(q, q_ref.detach(), 1e-4, msg="q")
    dxp = torch.empty(B, T, N, 64, device=dev)
    M = B * T * N
    grads = {k: torch.zeros_like(v) for k, v in pd.items()}
    ops.agent_unroll_bwd(w, cu(dq, dev), cu(dhs, dev), saved, hs, dxp, None,
                         {k: grads[k] for k in ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh",
                                                "fc2.weight", "fc2.bias")}, B, T, N, A)
    ops.linear_wgrad(dxp.view(M, 64), ops.src(obs_d.view(M, O), idx=u_d.view(M, 1), nhot=1, hot_w=A, nid=N),
                     grads["fc1.weight"], grads["fc1.bias"], M, 64, I)
    for k in p:
        ref = p[k].grad
        scale = max(1.0, float(ref.abs().max()))
        close(grads[k] / scale, ref / scale, 2e-4, 1e-3, msg=k)


@pytest.mark.parametrize("shape,B,T,pairs,with_dhs", [("2s3z", 7, 5, 1, False), ("2s3z", 40, 6, 2, True), ("2s3z", 700, 4, 1, False),
                                                       ("2s3z", 333, 3, 2, True), ("2s3z", 13, 9, 1, True), ("2s3z", 900, 3, 1, False),
                                                       ("2s3z", 1003, 4, 2, True), ("MMM2", 30, 4, 1, False), ("MMM2", 450, 3, 2, True),
                                                       ("3s5z", 61, 5, 2, True)])
def test_agent_unroll_bwd_x6_split(dev, shape, B, T, pairs, with_dhs):
    """BPTT on the bf16x6 split kernels (csrc/agent_bwd_x6.hip, opt-in gemm_mode): one or two sparse (action, value) pairs per row
    (the second pair with one value per (episode, step) shared by its agents, as QTRAN uses it), an optional external gradient on
    hs, one row tile per workgroup (up to 256 tiles) and two (beyond; row counts that leave the last workgroup with a partial or a
    missing second tile) - every gradient and dxp against
    torch autograd of the oracle unroll at the bounds of test_agent_unroll_bwd, beside the fp32 MFMA kernel on the same inputs."""
    from marl_amd import ops
    args, p_np, obs, ufed, h0 = _agent_case(shape, B, T, dev, seed=3)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions      # (MMM2: 18 actions = two action tiles of the fc2 gradient)
    I = O + A + N
    assert ops.agent_unroll_bwd_x6_supported(B, T, N, A)
    p, q_ref, hs_ref, _ = _oracle_unroll(args, p_np, obs, ufed, None, requires_grad=True)
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, A, (B, T, N), generator=g)
    gdiv = N if pairs == 2 else 1
    v1 = torch.randn(B, T, generator=g) if pairs == 2 else torch.randn(B, T, N, generator=g)
    e1 = v1[..., None].expand(B, T, N) if pairs == 2 else v1
    dense = torch.zeros(B, T, N, A).scatter_add_(3, idx[..., None], e1[..., None].contiguous())
    idx2 = v2 = None
    if pairs == 2:
        idx2 = torch.randint(0, A, (B, T, N), generator=g)
        idx2[0, 0] = idx[0, 0]                                # coinciding columns add
        v2 = torch.randn(B, T, generator=g)
        dense.scatter_add_(3, idx2[..., None], v2[..., None, None].expand(B, T, N, 1).contiguous())
    dhs = torch.randn(B, T, N, 64, generator=g) * 0.3 if with_dhs else None
    loss = (q_ref * dense).sum()
    if dhs is not None:
        loss = loss + (hs_ref * dhs).sum()
    loss.backward()
    pd = {k: cu(v, dev) for k, v in p_np.items()}
    w = ops.agent_weights(pd)
    q, hs = torch.empty(B, T, N, A, device=dev), torch.empty(B, T, N, 64, device=dev)
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
    obs_d, u_d = cu(obs, dev), cu(ufed, dev, torch.int32)
    ops.agent_unroll_fwd(w, obs_d, T * N, 0, u_d, T * N, 0, None, q, hs, None, saved, B, T, N, O, A)
    names = ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh", "fc2.weight", "fc2.bias")
    M = B * T * N
    out = {}
    for mode in ("x6", "f32"):
        grads = {k: torch.zeros_like(v) for k, v in pd.items()}
        dxp = torch.full((B, T, N, 64), float("nan"), device=dev)
        dh0 = torch.full((B * N, 64), float("nan"), device=dev)
        ops.agent_unroll_bwd(w, None, cu(dhs, dev) if dhs is not None else None, saved, hs, dxp, dh0, {k: grads[k] for k in names},
                             B, T, N, A, dq_idx=cu(idx, dev, torch.int32), dq_val=cu(v1, dev),
                             dq_idx2=cu(idx2, dev, torch.int32) if idx2 is not None else None,
                             dq_val2=cu(v2, dev) if v2 is not None else None, dq_gdiv=gdiv, x6=(mode == "x6"))
        ops.linear_wgrad(dxp.view(M, 64), ops.src(obs_d.view(M, O), idx=u_d.view(M, 1), nhot=1, hot_w=A, nid=N),
                         grads["fc1.weight"], grads["fc1.bias"], M, 64, I)
        out[mode] = (grads, dxp, dh0)
        for k in p:
            ref = p[k].grad
            scale = max(1.0, float(ref.abs().max()))
            close(grads[k] / scale, ref / scale, 2e-4, 1e-3, msg=mode + " " + k)
    close(out["x6"][1], out["f32"][1], 1e-5, 1e-4, msg="dxp")
    close(out["x6"][2], out["f32"][2], 1e-5, 1e-4, msg="dh0")
    worst = {m: max(float(((out[m][0][k].cpu() - p[k].grad).abs().max() / max(1.0, float(p[k].grad.abs().max())))) for k in p) for m in out}
    print("BPTT B=%d T=%d: worst scaled gradient error vs autograd: bf16x6 %.2e, fp32 MFMA %.2e" % (B, T, worst["x6"], worst["f32"]))


@pytest.mark.parametrize("shape,B,T", [("2s3z", 7, 5), ("MMM2", 3, 3), ("matrix", 9, 1), ("2s3z", 40, 6), ("2s3z", 3300, 3)])
def test_agent_unroll_bwd(dev, shape, B, T):
    """BPTT delta kernel + wgrad reductions vs torch autograd of the oracle unroll.  Up to four row tiles per workgroup a
    sparse dq runs the one-barrier pipelined kernel and a dense one the two-phase kernel (compared with each other below);
    B = 3300 (16 500 rows = five row tiles per workgroup, the headline layout) runs the two-phase kernel for both."""
    from marl_amd import ops
    args, p_np, obs, ufed, h0 = _agent_case(shape, B, T, dev, seed=1)
    N, O, A = args.n_agents, args.obs_shape, args.n_actions
    I = O + A + N
    p, q_ref, hs_ref, _ = _oracle_unroll(args, p_np, obs, ufed, None, requires_grad=True)
    g = torch.Generator().manual_seed(2)
    dq = torch.randn(B, T, N, A, generator=g)
    dhs = torch.randn(B, T, N, 64, generator=g) * 0.3
    ((q_ref * dq).sum() + (hs_ref * dhs).sum()).backward()

    pd = {k: cu(v, dev) for k, v in p_np.items()}
    w = ops.agent_weights(pd)
    q = torch.empty(B, T, N, A, device=dev)
    hs = torch.empty(B, T, N, 64, device=dev)
    saved = torch.empty(ops.saved_shape(T, B, N), device=dev)
    obs_d, u_d = cu(obs, dev), cu(ufed, dev, torch.int32)
    ops.agent_unroll_fwd(w, obs_d, T * N, 0, u_d, T * N, 0, None, q, hs, None, saved, B, T, N, O, A)
    dxp = torch.empty(B, T, N, 64, device=dev)
    dq_d = cu(dq, dev)
    M = B * T * N
    grads = {k: torch.zeros_like(v) for k, v in pd.items()}
    ops.agent_unroll_bwd(w, dq_d, cu(dhs, dev), saved, hs, dxp, None,
                         {k: grads[k] for k in ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh",
                                                "fc2.weight", "fc2.bias")}, B, T, N, A)
    ops.linear_wgrad(dxp.view(M, 64), ops.src(obs_d.view(M, O), idx=u_d.view(M, 1), nhot=1, hot_w=A, nid=N),
                     grads["fc1.weight"], grads["fc1.bias"], M, 64, I)
    for k in p:
        ref = p[k].grad
        scale = max(1.0, float(ref.abs().max()))
        close(grads[k] / scale, ref / scale, 2e-4, 1e-3, msg=k)
    # sparse form of dq (one (action, gradient) pair per row, no dhs) == the dense tensor it stands for
    idx = torch.randint(0, A, (B, T, N), generator=g)
    val = torch.randn(B, T, N, generator=g)
    dense = torch.zeros(B, T, N, A).scatter_(3, idx[..., None], val[..., None])
    names = ("rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh", "fc2.weight", "fc2.bias")
    ga = {k: torch.zeros_like(pd[k]) for k in names}
    gb = {k: torch.zeros_like(pd[k]) for k in names}
    dxa, dxb = torch.empty(B, T, N, 64, device=dev), torch.empty(B, T, N, 64, device=dev)
    ops.agent_unroll_bwd(w, cu(dense, dev), None, saved, hs, dxa, None, ga, B, T, N, A)
    ops.agent_unroll_bwd(w, None, None, saved, hs, dxb, None, gb, B, T, N, A,
                         dq_idx=cu(idx, dev, torch.int32), dq_val=cu(val, dev))
    close(dxb, dxa, 1e-6, 1e-5)
    for k in names:
        close(gb[k], ga[k], 1e-5, 1e-5, msg="sparse " + k)
    # two pairs per row with one value per (episode, step) shared by its agents (QTRAN: taken + greedy action, the sums
    # over agents' gradients; equal columns add) plus an external gradient on hs == the dense tensors they stand for
    idx2 = torch.randint(0, A, (B, T, N), generator=g)
    idx2[0, 0] = idx[0, 0]                                    # coinciding columns
    v1, v2 = torch.randn(B, T, generator=g), torch.randn(B, T, generator=g)
    dense2 = torch.zeros(B, T, N, A).scatter_add_(3, idx[..., None], v1[..., None, None].expand(B, T, N, 1).contiguous())
    dense2.scatter_add_(3, idx2[..., None], v2[..., None, None].expand(B, T, N, 1).contiguous())
    dhs2 = cu(torch.randn(B, T, N, 64, generator=g) * 0.1, dev)
    gc = {k: torch.zeros_like(pd[k]) for k in names}
    gd = {k: torch.zeros_like(pd[k]) for k in names}
    dxc, dxd = torch.empty(B, T, N, 64, device=dev), torch.empty(B, T, N, 64, device=dev)
    ops.agent_unroll_bwd(w, cu(dense2, dev), dhs2, saved, hs, dxc, None, gc, B, T, N, A)
    ops.agent_unroll_bwd(w, None, dhs2, saved, hs, dxd, None, gd, B, T, N, A, dq_idx=cu(idx, dev, torch.int32),
                         dq_val=cu(v1, dev), dq_idx2=cu(idx2, dev, torch.int32), dq_val2=cu(v2, dev), dq_gdiv=N)
    close(dxd, dxc, 1e-6, 1e-5)
    for k in names:
        close(gd[k], gc[k], 1e-5, 1e-5, msg="two-pair sparse " + k)


# ------------------------------------------------------------------------------------- per-row
def test_select_kernels(dev):
    from marl_amd import ops
    g = torch.Generator().manual_seed(4)
    R, A = 1000, 11
    q = torch.randn(R, A, generator=g)
    q[5, 3] = q[5, 7] = 9.0        # tie -> first index
    avail = (torch.rand(R, A, generator=g) < 0.7).float()
    avail[:, 0] = 1
    idx = torch.randint(0, A, (R,), generator=g)
    qd, ad, idd = cu(q, dev), cu(avail, dev), cu(idx, dev, torch.int32)
    out = torch.empty(R, device=dev)
    ops.q_gather(qd, idd, out, R, A)
    close(out, q.gather(1, idx[:, None]).squeeze(1), 0, 0)
    qm = q.clone(); qm[avail == 0] = -9999999.0
    mx, am = torch.empty(R, device=dev), torch.empty(R, dtype=torch.int32, device=dev)
    ops.q_masked_max(qd, ad, -9999999.0, mx, am, R, A)
    close(mx, qm.max(1)[0], 0, 0)
    assert (am.cpu().long() == qm.argmax(1)).all()
    # >= 4096 rows with 16-byte aligned operands: the LDS-staged kernel (ragged last tile, rows with nothing available, ties,
    # no availability mask, 14 / 18 actions); an operand off the 16-byte grid keeps the row-per-thread kernel
    for RR, AA in ((5003, 14), (4096, 18), (70001, 11)):
        qq = torch.randn(RR, AA, generator=g)
        av = (torch.rand(RR, AA, generator=g) < 0.6).float()
        av[::9] = 0
        qq[7, 2] = qq[7, AA - 1] = 11.0; av[7] = 1
        for use_av in (True, False):
            ref = qq.clone()
            if use_av:
                ref[av == 0] = -9999999.0
            mx2, am2 = torch.empty(RR, device=dev), torch.empty(RR, dtype=torch.int32, device=dev)
            ops.q_masked_max(cu(qq, dev), cu(av, dev) if use_av else None, -9999999.0, mx2, am2, RR, AA)
            close(mx2, ref.max(1)[0], 0, 0)
            assert (am2.cpu().long() == ref.argmax(1)).all()
        un = torch.empty(RR * AA + 1, device=dev)
        un[1:] = cu(qq, dev).reshape(-1)
        mx3, am3 = torch.empty(RR, device=dev), torch.empty(RR, dtype=torch.int32, device=dev)
        ops.q_masked_max(un[1:].view(RR, AA), cu(av, dev), -9999999.0, mx3, am3, RR, AA)
        ref = qq.clone(); ref[av == 0] = -9999999.0
        close(mx3, ref.max(1)[0], 0, 0)
        assert (am3.cpu().long() == ref.argmax(1)).all()
    # fused double-Q selection: argmax of the masked eval-next Q, target Q gathered there (q_learner.py:104-117)
    for RR, AA in ((R, A), (4099, 18), (63, 3)):
        qs, qv = torch.randn(RR, AA, generator=g), torch.randn(RR, AA, generator=g)
        av = (torch.rand(RR, AA, generator=g) < 0.6).float()
        av[::7] = 0                                    # rows with nothing available
        qs[3, 1] = qs[3, AA - 1] = 5.0; av[3] = 1      # tie -> first index
        qsm, qvm = qs.clone(), qv.clone()
        qsm[av == 0] = -9999999.0; qvm[av == 0] = -9999999.0
        ref_arg = qsm.argmax(1)
        ov, oa = torch.empty(RR, device=dev), torch.empty(RR, dtype=torch.int32, device=dev)
        ops.q_double_select(cu(qs, dev), cu(qv, dev), cu(av, dev), -9999999.0, ov, oa, RR, AA)
        assert (oa.cpu().long() == ref_arg).all()
        un = torch.empty(RR * AA + 1, device=dev)           # operand at an address that is not 16-byte aligned
        un[1:] = cu(qs, dev).reshape(-1)
        ov2, oa2 = torch.empty(RR, device=dev), torch.empty(RR, dtype=torch.int32, device=dev)
        ops.q_double_select(un[1:].view(RR, AA), cu(qv, dev), cu(av, dev), -9999999.0, ov2, oa2, RR, AA)
        assert torch.equal(oa2, oa) and torch.equal(ov2, ov)
        close(ov, qvm.gather(1, ref_arg[:, None]).squeeze(1), 0, 0)
    g1, g2 = torch.randn(R // 5, generator=g), torch.randn(R // 5, generator=g)
    dq = torch.empty(R, A, device=dev)
    ops.q_scatter(dq, idd, cu(g1, dev), am, cu(g2, dev), R, A, gdiv=5)
    ref = torch.zeros(R, A)
    ref[torch.arange(R), idx] += g1.repeat_interleave(5)
    ref[torch.arange(R), qm.argmax(1)] += g2.repeat_interleave(5)
    close(dq, ref, 1e-6)
    x = torch.randn(40, 5, 7, generator=g)
    s = torch.empty(40, 7, device=dev)
    ops.agent_sum(cu(x, dev), s, 40, 5, 7)
    close(s, x.sum(1), 1e-5)
    bc = cu(x, dev)
    ops.agent_bcast(s, bc, 40, 5, 7, accumulate=True)
    close(bc, x + x.sum(1, keepdim=True), 1e-5)


def test_qmix_mix(dev):
    from marl_amd import ops
    g = torch.Generator().manual_seed(6)
    R, N, E = 333, 5, 32
    Wd = N * E + 3 * E
    hy = torch.randn(R, Wd, generator=g, requires_grad=True)
    b2 = torch.randn(R, generator=g, requires_grad=True)
    q = torch.randn(R, N, generator=g, requires_grad=True)
    w1 = hy[:, :N * E].abs().view(R, N, E)
    hid = F.elu((q.unsqueeze(2) * w1).sum(1) + hy[:, N * E:N * E + E])
    qt = (hid * hy[:, N * E + E:N * E + 2 * E].abs()).sum(1) + b2
    gq = torch.randn(R, generator=g)
    (qt * gq).sum().backward()
    hyd, qd = cu(hy.detach(), dev), cu(q.detach(), dev)
    out = torch.empty(R, device=dev)
    ops.qmix_mix_fwd(hyd, cu(b2.detach(), dev), qd, out, R, N, E)
    close(out, qt, 1e-4)
    dhy = torch.zeros(R, Wd, device=dev)
    db2, dq = torch.empty(R, device=dev), torch.empty(R, N, device=dev)
    ops.qmix_mix_bwd(hyd, qd, cu(gq, dev), dhy, db2, dq, R, N, E)
    close(dhy[:, :N * E + 2 * E], hy.grad[:, :N * E + 2 * E], 1e-4)
    close(db2, b2.grad, 1e-6)
    close(dq, q.grad, 1e-4)


@pytest.mark.parametrize("x6", [False, True], ids=["f32", "bf16x6"])
@pytest.mark.parametrize("R,N,S", [(333, 5, 120), (16, 5, 120), (4099, 3, 48), (50, 2, 4), (1000, 5, 126), (257, 4, 128), (100, 1, 128),
                                   (8200, 5, 124)])
def test_qmix_fused(dev, R, N, S, x6):
    """fused hypernet + mixing kernels (forward, dq, hypernet weight gradients; fp32 MFMA and the bf16x6 split variant) vs torch-CPU
    autograd of the restated QMixMixer.forward (reference network/mixer.py:57-80); S not a multiple of 4 falls back."""
    from marl_amd import ops
    E = 32
    assert ops.qmix_fused_supported(N, S, E)
    g = torch.Generator().manual_seed(R + N + S)
    names = ("w1", "b1", "w2", "h")
    outs = {"w1": N * E, "b1": E, "w2": E, "h": E}
    P = {}
    for k in names:
        P[k] = (torch.randn(outs[k], S, generator=g) * 0.2).requires_grad_()
        P[k + "_b"] = (torch.randn(outs[k], generator=g) * 0.2).requires_grad_()
    P["b2_w"] = torch.randn(1, E, generator=g).requires_grad_()
    P["b2_b"] = torch.randn(1, generator=g).requires_grad_()
    s = torch.randn(R, S, generator=g)
    q = torch.randn(R, N, generator=g, requires_grad=True)
    gq = torch.randn(R, generator=g)
    w1 = F.linear(s, P["w1"], P["w1_b"]).abs().view(R, N, E)
    hid = F.elu((q.unsqueeze(2) * w1).sum(1) + F.linear(s, P["b1"], P["b1_b"]))
    qt = (hid * F.linear(s, P["w2"], P["w2_b"]).abs()).sum(1) + \
        F.linear(torch.relu(F.linear(s, P["h"], P["h_b"])), P["b2_w"], P["b2_b"]).squeeze(1)
    (qt * gq).sum().backward()
    Wd = {k: cu(v.detach(), dev) for k, v in P.items()}
    base = {k: torch.randn(v.shape, generator=g) for k, v in P.items()}      # gradients accumulate
    Gd = {k: cu(v, dev) for k, v in base.items()}
    ld = (S + 3) // 4 * 4 + 4                                                # padded row stride
    sd = torch.zeros(R, ld, device=dev)
    sd[:, :S] = cu(s, dev)
    if S % 4:
        return                                                               # generic composition handles it
    xs = ops.src(sd[:, :S])
    out = torch.full((R,), 9.0, device=dev)
    qd = cu(q.detach(), dev)
    ops.qmix_fused_fwd(ops.qmix_weights(Wd), xs, qd, out, R, N, S, E, x6=x6)
    close(out, qt, 1e-4)
    dq = torch.full((R, N), 9.0, device=dev)
    ops.qmix_fused_bwd(ops.qmix_weights(Wd), xs, qd, cu(gq, dev), dq, ops.qmix_weights(Gd), R, N, S, E, x6=x6)
    close(dq, q.grad, 1e-4)
    scale = max(1.0, (R / 64.0) ** 0.5)
    for k, v in P.items():
        close(Gd[k], base[k] + v.grad, 2e-4 * scale, 1e-4, msg=k)


def test_qplex_mix(dev):
    from marl_amd import ops
    g = torch.Generator().manual_seed(7)
    R, N, K = 257, 5, 10
    t = lambda *s: torch.randn(*s, generator=g, requires_grad=True)
    w_raw, v, q, key, ag, ac = t(R, N), t(R, N), t(R, N), t(R, K), t(R, K, N), t(R, K, N)
    mx = q.detach() + torch.rand(R, N, generator=g)
    w = w_raw.abs() + 1e-10
    qt = w * q + v
    mt = w * mx + v
    lam = ((key.abs() + 1e-10).unsqueeze(2) * torch.sigmoid(ag) * torch.sigmoid(ac)).sum(1)
    v_tot = qt.sum(1)
    a_tot = ((qt - mt).detach() * (lam - 1)).sum(1)
    gq = torch.randn(R, generator=g)
    ((v_tot + a_tot) * gq).sum().backward()
    d = lambda x: cu(x.detach(), dev)
    vt, at, lo = torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, N, device=dev)
    ops.qplex_mix_fwd(d(w_raw), d(v), d(q), d(mx), d(key), d(ag), d(ac), vt, at, lo, R, N, K, True, True)
    close(vt, v_tot, 1e-4); close(at, a_tot, 1e-4); close(lo, lam, 1e-4)
    outs = [torch.empty_like(d(x)) for x in (q, w_raw, v, key, ag, ac)]
    ops.qplex_mix_bwd(d(w_raw), d(q), d(mx), d(key), d(ag), d(ac), cu(gq, dev), *outs, R, N, K, True, True)
    for o, ref in zip(outs, (q, w_raw, v, key, ag, ac)):
        close(o, ref.grad, 1e-4)


def test_losses_and_optimizer(dev):
    from marl_amd import ops
    g = torch.Generator().manual_seed(8)
    R = 5000
    qt, qg, r = torch.randn(R, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g)
    term = (torch.rand(R, generator=g) < 0.1).float()
    pad = (torch.rand(R, generator=g) < 0.2).float()
    mask = 1 - pad
    td = (r + 0.99 * qg * (1 - term)) - qt
    out2, dqt = torch.empty(2, device=dev), torch.empty(R, device=dev)
    ops.td_loss(cu(qt, dev), cu(qg, dev), cu(r, dev), cu(term, dev), cu(pad, dev), 0.99, dqt, out2, R)
    close(out2, torch.stack([((mask * td) ** 2).sum(), mask.sum()]), rtol=1e-5, atol=1e-3)
    close(dqt, -2 * mask * td, 1e-5)
    # qtran
    jq, jt, v, jh, so, sn = [torch.randn(R, generator=g) for _ in range(6)]
    jq.requires_grad_(True); v.requires_grad_(True); so.requires_grad_(True); sn.requires_grad_(True)
    y = r + 0.99 * jt * (1 - term)
    l_td = (((jq - y) * mask) ** 2).sum()
    l_opt = (((so - jh + v) * mask) ** 2).sum()
    l_nopt = (((sn - jq.detach() + v).clamp(max=0) * mask) ** 2).sum()
    (l_td + 1.0 * l_opt + 1.0 * l_nopt).backward()
    d = lambda x: cu(x.detach(), dev)
    outs = [torch.empty(R, device=dev) for _ in range(4)]
    out4 = torch.empty(4, device=dev)
    ops.qtran_loss(d(jq), d(jt), d(v), d(jh), d(so), d(sn), cu(r, dev), cu(term, dev), cu(pad, dev), 0.99, 1.0, 1.0,
                   *outs, out4, R)
    close(out4, torch.stack([l_td, l_opt, l_nopt, mask.sum()]).detach(), rtol=1e-5, atol=1e-2)
    for o, ref in zip(outs, (jq, v, so, sn)):
        close(o, ref.grad, 1e-5)
    # optimizer: clip + RMSprop / Adam against torch.optim on the normalised gradient
    n = 70001
    for kind in ("RMS", "Adam"):
        p0 = torch.randn(n, generator=g)
        pt = torch.nn.Parameter(p0.clone())
        opt = torch.optim.RMSprop([pt], lr=5e-4) if kind == "RMS" else torch.optim.Adam([pt], lr=5e-4)
        pd = cu(p0, dev)
        s1, s2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        den = cu(np.array([37.0]), dev)
        sumsq = torch.empty(1, device=dev)
        for step in range(1, 4):
            graw = torch.randn(n, generator=g) * (50.0 if step == 1 else 0.5)
            pt.grad = graw / 37.0
            torch.nn.utils.clip_grad_norm_([pt], 10)
            opt.step()
            gd = cu(graw, dev)
            ops.grad_sumsq(gd, n, sumsq)
            if kind == "RMS":
                ops.rmsprop_step(pd, gd, s1, n, 5e-4, 0.99, 1e-8, 10, sumsq, den)
            else:
                ops.adam_step(pd, gd, s1, s2, n, 5e-4, 0.9, 0.999, 1e-8, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5,
                              10, sumsq, den)
            close(pd, pt.detach(), 2e-6, 1e-5, msg="%s step %d" % (kind, step))


def test_rollout_kernels_match_numpy_env(dev):
    from marl_amd import ops
    E, T, N, O, S, A = 9, 6, 5, 80, 120, 11
    sy = orl.SynthSMAC(N, O, S, A, T, seed=5)
    env = np.arange(3, 3 + E)
    length = torch.empty(E, dtype=torch.int32, device=dev)
    won = torch.empty(E, dtype=torch.int32, device=dev)
    ops.synth_lengths(5, 3, 2, length, won, E, T)
    L = sy.length(env, np.full(E, 2))
    assert (length.cpu().numpy() == L).all()
    assert (won.cpu().numpy().astype(bool) == sy.won(env, np.full(E, 2))).all()
    obs = torch.empty(E, T + 1, N, O, device=dev); st = torch.empty(E, T + 1, S, device=dev)
    av = torch.empty(E, T + 1, N, A, device=dev)
    u = torch.empty(E, T, N, dtype=torch.int32, device=dev)
    r, term, pad = (torch.empty(E, T, device=dev) for _ in range(3))
    alive = torch.ones(E, dtype=torch.int32, device=dev)
    act = torch.empty(E, N, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(0)
    for t in range(T + 1):
        ops.synth_observe(5, 3, 2, t, length, obs, st, av, E, T, N, O, S, A)
        live = (t <= L)
        ro = sy.obs(env, np.full(E, 2), t) * live[:, None, None]
        np.testing.assert_array_equal(obs[:, t].cpu().numpy(), ro.astype(np.float32))
        np.testing.assert_array_equal(st[:, t].cpu().numpy(), (sy.state(env, np.full(E, 2), t) * live[:, None]).astype(np.float32))
        ra = sy.avail(env, np.full(E, 2), t) * live[:, None, None]
        np.testing.assert_array_equal(av[:, t].cpu().numpy(), ra.astype(np.float32))
        if t == T:
            break
        # epsilon-greedy selection vs the numpy formula
        q = rng.standard_normal((E, N, A)).astype(np.float32)
        ops.select_actions(cu(q, dev), av[:, t], (T + 1) * N * A, alive, 0.4, 77, 3, None, 2 * (T + 1) + t, act, N, E, N, A)
        a_t = ra
        qm = q.copy(); qm[a_t == 0] = -np.inf
        tg = np.full((E, 1), 2 * (T + 1) + t)
        explore = orl.u01(orl.key(77, orl.ST_EXPLORE, env[:, None], tg, np.arange(N)[None])) < np.float32(0.4)
        nav = a_t.sum(-1).astype(np.int64)
        k = np.floor(orl.u01(orl.key(77, orl.ST_PICK, env[:, None], tg, np.arange(N)[None])) * nav.astype(np.float32)).astype(np.int64)
        k = np.minimum(k, np.maximum(nav - 1, 0))
        pick = (np.cumsum(a_t, -1) <= k[..., None]).sum(-1)
        ref_act = np.where(explore, pick, qm.argmax(-1))
        got = act.cpu().numpy()
        al = alive.cpu().numpy().astype(bool)
        assert (got[al] == ref_act[al]).all()
        assert (got[~al] == -1).all()
        ops.synth_step(5, 3, 2, t, length, act, u, r, term, pad, alive, E, T, N, A)
        rr = sy.reward(env, np.full(E, 2), t, np.where(got < 0, 0, got))
        livet = t < L
        np.testing.assert_array_equal(r[:, t].cpu().numpy(), np.where(livet, rr, 0).astype(np.float32))
        np.testing.assert_array_equal(pad[:, t].cpu().numpy(), (~livet).astype(np.float32))
        np.testing.assert_array_equal(term[:, t].cpu().numpy(), np.where(livet, (t + 1 >= L), 1).astype(np.float32))
        assert (alive.cpu().numpy() == (t + 1 < L)).all()


@pytest.mark.parametrize("kind,BT,N,A,S", [("q", 37, 8, 14, 216), ("q", 16, 5, 11, 120), ("q", 1, 2, 3, 1), ("q", 300, 3, 16, 40),
                                            ("v", 37, 8, 14, 216), ("v", 129, 5, 11, 120), ("q", 4100, 8, 14, 216),
                                            ("q", 700, 3, 9, 320), ("v", 333, 4, 5, 228), ("v", 9000, 2, 5, 224), ("q", 64, 2, 16, 384)])
def test_qtran_fused_heads(dev, kind, BT, N, A, S):
    """Fused QtranQBase / QtranV kernels (csrc/qtran_fused.hip) behind the mixer classes: forward, gradient on the
    hidden states and every parameter gradient vs torch-CPU autograd of the reference forward (network/mixer.py:378-388,
    :411-418: per-agent encoder, THEN the agent sum), plus equality with the generic marl_linear composition."""
    import types
    from marl_amd.network.mixer import QtranQBase, QtranV
    from marl_amd.hostutil import FlatParams
    args = types.SimpleNamespace(n_agents=N, n_actions=A, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
    torch.manual_seed(BT + N + A)
    mod = (QtranQBase if kind == "q" else QtranV)(args)
    ref = (QtranQBase if kind == "q" else QtranV)(args)
    ref.load_state_dict(mod.state_dict())
    g = torch.Generator().manual_seed(7 * BT + N)
    R = BT * N
    s = torch.randn(BT, S, generator=g)
    h = (torch.randn(R, 64, generator=g) * 0.7).requires_grad_()
    u = torch.randint(0, A, (R,), generator=g).int()
    u[::7] = -1                                              # padding rows: all-zero one-hot
    d_out = torch.randn(BT, generator=g)
    # torch-CPU reference (parameters of `ref` get .grad)
    enc = ref.hidden_action_encoding if kind == "q" else ref.hidden_encoding
    head = ref.q if kind == "q" else ref.v
    if kind == "q":
        oh = torch.zeros(R, A)
        oh[u >= 0] = F.one_hot(u[u >= 0].long(), A).float()
        x = torch.cat([h, oh], dim=1)
    else:
        x = h
    esum = enc(x).view(BT, N, -1).sum(1)
    out_ref = head(torch.cat([s, esum], dim=1)).squeeze(1)
    e1_pre = enc[0](x).detach()                               # (R, AE) pre-activations of the encoder's relu

    def close_except_kinks(got, msg):
        """The gradient wrt hidden is discontinuous where an encoder pre-activation crosses 0: a unit whose
        pre-activation is within fp32 rounding of 0 may be gated differently by two correct summation orders (the
        kernel adds the one-hot column as a table bias, torch inside the GEMM).  Rows are compared at 2e-5 / 1e-4;
        a row may only deviate if it HAS such a unit (|pre-activation| < 2e-6), and only a handful of rows may."""
        a_, b_ = got.detach().cpu().numpy(), h.grad.numpy()
        bad = np.unique(np.nonzero(np.abs(a_ - b_) > 2e-5 + 1e-4 * np.abs(b_))[0])
        assert len(bad) <= max(1, R // 10000), (msg, len(bad))
        for r in bad:
            assert float(e1_pre[r].abs().min()) < 2e-6, (msg, int(r), float(e1_pre[r].abs().min()))
    (out_ref * d_out).sum().backward()
    # product: parameters in one flat buffer with gradient views (as in the learners)
    mod.to(dev)
    fp = FlatParams(list(mod.parameters()), dev, with_grad=True)
    base = torch.randn(fp.n, generator=g).to(dev)            # gradients accumulate into what is there
    fp.grad.copy_(base)
    sd, hd, ud, dd = cu(s, dev), cu(h.detach(), dev), cu(u, dev, torch.int32), cu(d_out, dev)
    assert mod._qt_ok(hd)
    dh = {}
    for fused in (True, False):
        mod.no_fused = not fused
        fp.grad.copy_(base)
        ctx = {}
        out = mod.hip_forward(sd, hd, ud, BT, ctx=ctx) if kind == "q" else mod.hip_forward(sd, hd, BT, ctx=ctx)
        assert bool(ctx.get("fused")) == fused
        close(out, out_ref, 2e-5, 1e-4, msg="out fused=%s" % fused)
        dhid = torch.full((R, 64), 0.5, device=dev)
        mod.hip_backward(ctx, dd, BT, dhid, accumulate=True)
        close_except_kinks(dhid - 0.5, "dhidden fused=%s" % fused)
        dh[fused] = dhid.clone()
        scale = max(1.0, (R / 64.0) ** 0.5)
        for (name, p), pr in zip(mod.named_parameters(), ref.parameters()):
            want = pr.grad
            got = p.grad - base[fp.offsets[[id(q) for q in fp.params].index(id(p))]:][:p.numel()].view(p.shape)
            tol = 3e-5 * scale * max(1.0, float(want.abs().max()))
            close(got, want, tol, 1e-4, msg="%s fused=%s" % (name, fused))
    # without accumulate the old contents are overwritten
    mod.no_fused = False
    ctx = {}
    out = mod.hip_forward(sd, hd, ud, BT, ctx=ctx) if kind == "q" else mod.hip_forward(sd, hd, BT, ctx=ctx)
    dhid = torch.full((R, 64), 123.0, device=dev)
    mod.hip_backward(ctx, dd, BT, dhid, accumulate=False)
    close_except_kinks(dhid, "dhidden overwrite")
    # the public forward() (reference signature) takes the fused path too
    if kind == "q":
        oh4 = torch.zeros(R, A); oh4[u >= 0] = F.one_hot(u[u >= 0].long(), A).float()
        pub = mod(s.view(1, BT, S), h.detach().view(1, BT, N, 64), oh4.view(1, BT, N, A))
    else:
        pub = mod(s.view(1, BT, S), h.detach().view(1, BT, N, 64))
    close(pub.view(-1), out_ref, 2e-5, 1e-4, msg="public forward")


@pytest.mark.parametrize("BT,S", [(37, 216), (4100, 216), (64, 4), (1000, 384), (129, 120)])
def test_qtran_state_parts(dev, BT, S):
    """marl_qtran_state_parts behind _QtranFusedHead.state_part: the state columns of q.0 / v.0 (network/mixer.py:386, :416)
    for one head and for the joint-Q / V pair in one pass, vs torch fp64."""
    import types
    from marl_amd.network.mixer import QtranQBase, QtranV
    from marl_amd import ops
    assert ops.qtran_state_parts_supported(S)
    args = types.SimpleNamespace(n_agents=3, n_actions=7, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
    torch.manual_seed(BT + S)
    qn, vn = QtranQBase(args).to(dev), QtranV(args).to(dev)
    s = torch.randn(BT, S, generator=torch.Generator().manual_seed(S)).to(dev)

    def want(net):
        l0 = net._qt_layers()[2]
        return (s.double() @ l0.weight.data[:, :S].double().t() + l0.bias.data.double()).float()
    one = qn.state_part(s, BT, "x").clone()
    close(one, want(qn), 2e-5, 1e-4, msg="single")
    a, b = qn.state_part(s, BT, "y", other=vn)
    close(a, want(qn), 2e-5, 1e-4, msg="pair/q")
    close(b, want(vn), 2e-5, 1e-4, msg="pair/v")
    assert torch.equal(a, one)                                # the pair kernel multiplies in the same order


@pytest.mark.parametrize("BT,S", [(100, 322), (70, 50), (33, 121)])
def test_qtran_state_parts_ignores_what_follows_a_row(dev, BT, S):
    """S % 4 != 0: the kernel reads a row's last 16 bytes, which reach past S into whatever the storage holds there - a column
    slice of a wider buffer may carry NaN / Inf neighbours.  They are zeroed before the multiply."""
    import types
    from marl_amd.network.mixer import QtranQBase
    from marl_amd import ops
    if not ops.qtran_state_parts_supported(S):
        pytest.skip("state width not covered by the row kernel")
    args = types.SimpleNamespace(n_agents=3, n_actions=7, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
    torch.manual_seed(BT + S)
    qn = QtranQBase(args).to(dev)
    SP = (S + 3) // 4 * 4
    wide = torch.randn(BT, SP, generator=torch.Generator().manual_seed(S)).to(dev)
    wide[:, S:] = float("nan")
    wide[::2, S:] = float("inf")
    s = wide[:, :S]
    assert ops.qtran_state_parts_supported(S, s)
    l0 = qn._qt_layers()[2]
    want = (s.double() @ l0.weight.data[:, :S].double().t() + l0.bias.data.double()).float()
    got = qn.state_part(s, BT, "x").clone()
    assert torch.isfinite(got).all()
    close(got, want, 2e-5, 1e-4, msg="slice with non-finite neighbours")


@pytest.mark.parametrize("rows,N,S,padded", [(37, 5, 120, False), (2500, 10, 322, True), (130, 3, 48, True)])
def test_qmix_two_hyper_tail(dev, rows, N, S, padded):
    """QMixMixer(two_hyper_layers) generic path: hyper_b1 / hyper_b2.0 from one pass over s (marl_qmix_tail_fwd) and hyper_b2.2 inside
    the mixing kernels (network/mixer.py:44-47, :69-77) vs the marl_linear composition and vs torch autograd of the reference
    forward; `padded`: the states are rows of 16-byte padded storage (S % 4 != 0 on MMM2: S = 322 in 324-float rows)."""
    import types
    from marl_amd.network.mixer import QMixMixer
    from marl_amd.hostutil import FlatParams
    from marl_amd import ops
    args = types.SimpleNamespace(n_agents=N, state_shape=S, qmix_hidden_dim=32, hyper_hidden_dim=64, two_hyper_layers=True)
    torch.manual_seed(rows + S)
    mod, ref = QMixMixer(args), QMixMixer(args)
    ref.load_state_dict(mod.state_dict())
    g = torch.Generator().manual_seed(rows)
    s = torch.randn(rows, S, generator=g)
    q = torch.randn(rows, N, generator=g).requires_grad_()
    gq = torch.randn(rows, generator=g)
    # torch reference (mixer.py:57-80)
    w1 = torch.abs(ref.hyper_w1(s)).view(rows, N, 32)
    b1 = ref.hyper_b1(s).view(rows, 1, 32)
    hid = F.elu(torch.bmm(q.view(rows, 1, N), w1) + b1)
    w2 = torch.abs(ref.hyper_w2(s)).view(rows, 32, 1)
    want = (torch.bmm(hid, w2) + ref.hyper_b2(s).view(rows, 1, 1)).view(rows)
    (want * gq).sum().backward()
    mod.to(dev)
    fp = FlatParams(list(mod.parameters()), dev, with_grad=True)
    if padded:
        ld = (S + 3) // 4 * 4
        store = torch.zeros(rows, ld, device=dev)
        store[:, :S] = s.to(dev)
        sd = store[:, :S]                                        # unit inner stride, 16-byte aligned rows
    else:
        sd = s.to(dev)
    qd, gd = q.detach().to(dev), gq.to(dev)
    outs = {}
    for tail in (True, False):
        mod.no_fused = not tail
        fp.grad.zero_()
        ctx = {}
        qt = mod.hip_forward(qd, sd, rows, ctx=ctx).clone()
        assert bool(ctx.get("tail")) == tail
        dq = mod.hip_backward(ctx, gd, rows).clone()
        outs[tail] = (qt, dq, fp.grad.clone())
        close(qt, want, 2e-5, 1e-4, msg="q_tot tail=%s" % tail)
        close(dq, q.grad, 2e-5, 1e-4, msg="dq tail=%s" % tail)
        scale = max(1.0, (rows / 64.0) ** 0.5)
        for (name, p_), pr in zip(mod.named_parameters(), ref.parameters()):
            close(p_.grad, pr.grad, 3e-5 * scale * max(1.0, float(pr.grad.abs().max())), 1e-4, msg="%s tail=%s" % (name, tail))


@pytest.mark.parametrize("BT,N,A,S", [(37, 8, 14, 216), (4100, 3, 16, 40), (1, 2, 3, 1)])
def test_qtran_two_action_sets(dev, BT, N, A, S):
    """QtranQBase.hip_forward(u_idx2=...): the joint-Q head for the taken and the greedy actions in one launch
    (qtran_learner.py:116, :133).  The first set's outputs and saved activations are the bits of the single call; the second
    set's value matches its own single call within fp32 rounding (its pre-activation is formed as first + table difference)."""
    import types
    from marl_amd.network.mixer import QtranQBase
    args = types.SimpleNamespace(n_agents=N, n_actions=A, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
    torch.manual_seed(BT + A)
    mod = QtranQBase(args).to(dev)
    g = torch.Generator().manual_seed(BT)
    s = torch.randn(BT, S, generator=g).to(dev)
    h = (torch.randn(BT * N, 64, generator=g) * 0.7).to(dev)
    u1 = torch.randint(0, A, (BT * N,), generator=g).int()
    u1[::5] = -1
    u2 = torch.randint(0, A, (BT * N,), generator=g).int()
    u1, u2 = u1.to(dev), u2.to(dev)
    assert mod._qt_ok(h)
    c1, c2 = {}, {}
    a1 = mod.hip_forward(s, h, u1, BT, ctx=c1, tag="a").clone()
    saved1 = {k: c1[k].clone() for k in ("s1", "e2", "y1", "y2")}
    b1 = mod.hip_forward(s, h, u2, BT, tag="b").clone()
    a2, b2 = mod.hip_forward(s, h, u1, BT, ctx=c2, tag="c", u_idx2=u2)
    assert torch.equal(a1, a2)
    for k, v in saved1.items():
        assert torch.equal(v, c2[k]), k
    close(b2, b1, 2e-5, 1e-4, msg="second action set")


@pytest.mark.parametrize("kind,BT,S", [("q", 37, 216), ("q", 8229, 216), ("v", 2500, 384)])
def test_qtran_row_kernels_reproducible_and_remapped(dev, kind, BT, S):
    """The row-level kernels (state parts, weight gradients) on states read in place from (T+1)-slot storage through a row
    remap and an episode map give the bits of the dense call, and repeated calls give the same bits (slabs, fixed-order
    reduce)."""
    import types
    from marl_amd.network.mixer import QtranQBase, QtranV
    from marl_amd.hostutil import FlatParams
    from marl_amd import ops
    N, A, T = 3, 6, 11
    B = (BT + T - 1) // T
    BT = B * T
    args = types.SimpleNamespace(n_agents=N, n_actions=A, state_shape=S, rnn_hidden_dim=64, qtran_hidden_dim=64)
    torch.manual_seed(BT)
    mod = (QtranQBase if kind == "q" else QtranV)(args).to(dev)
    fp = FlatParams(list(mod.parameters()), dev, with_grad=True)
    g = torch.Generator().manual_seed(BT + 1)
    store = torch.randn(B + 5, T + 1, S, generator=g).to(dev)              # episode storage, T + 1 slots
    emap = torch.randperm(B + 5, generator=g)[:B].int().to(dev)
    rows = ops.Rows(store.view(-1, S), (T, T + 1, 1), emap)                 # s_next of the sampled episodes
    dense = store[emap.long(), 1:].reshape(BT, S).contiguous()
    h = (torch.randn(BT * N, 64, generator=g) * 0.7).to(dev)
    u = torch.randint(0, A, (BT * N,), generator=g).int().to(dev)
    d = torch.randn(BT, generator=g).to(dev)
    got = []
    for src in (dense, rows, rows, dense, rows):
        fp.grad.zero_()
        ctx = {}
        sp = mod.state_part(src, BT, "e").clone()
        out = (mod.hip_forward(src, h, u, BT, ctx=ctx, sp=sp) if kind == "q" else mod.hip_forward(src, h, BT, ctx=ctx, sp=sp)).clone()
        assert ctx.get("fused")
        dh = torch.zeros(BT * N, 64, device=dev)
        mod.hip_backward(ctx, d, BT, dh, accumulate=False)
        got.append((sp, out, fp.grad.clone()))
    for other in got[1:]:
        for a_, b_ in zip(got[0], other):
            assert torch.equal(a_, b_)
    assert float(got[0][2].abs().max()) > 0


def _qmix_reference(P, s, q, gq, N, E, bf16, wgrad_fp32=False):
    """QMixMixer.forward (reference network/mixer.py:57-80) on torch-CPU; bf16 = True rounds BOTH operands of the four
    state-conditioned hypernet GEMMs to bf16 (fp32 accumulation), which is what the bf16 matrix-core path computes.
    wgrad_fp32: same forward values, but the weight gradient is d(out)^T s with the UNROUNDED states (what the C-ABI computes
    with flags = 1, i.e. without the bf16 weight-gradient bit)"""
    rnd = (lambda t: t.bfloat16().float()) if bf16 else (lambda t: t)
    R = s.shape[0]
    if bf16 and wgrad_fp32:
        def lin(k):
            plain = F.linear(s, P[k])
            return F.linear(rnd(s), rnd(P[k].detach()), P[k + "_b"]) + (plain - plain.detach())      # + exactly 0, gradient d(out)^T s
    else:
        lin = lambda k: F.linear(rnd(s), rnd(P[k]), P[k + "_b"])
    w1 = lin("w1").abs().view(R, N, E)
    hid = F.elu((q.unsqueeze(2) * w1).sum(1) + lin("b1"))
    qt = (hid * lin("w2").abs()).sum(1) + F.linear(torch.relu(lin("h")), P["b2_w"], P["b2_b"]).squeeze(1)
    (qt * gq).sum().backward()
    return qt


@pytest.mark.parametrize("R,remap", [(40007, False), (32768, False), (36000, True)])
def test_qmix_wide_resident_forward(dev, R, remap):
    """bf16 forward with the weights resident in LDS (qmix_wide_res_fwd_kernel: MMM2 shape, >= 32 768 rows; each row's q_tot is
    the sum of two embedding halves computed by two workgroups) vs torch-CPU with the hypernet operands rounded to bf16,
    and vs the streaming kernel (experiments wide_res = 0) - the two differ only in the order of the final sums."""
    import os
    from marl_amd import ops
    N, S, E = 10, 322, 32
    g = torch.Generator().manual_seed(R)
    outs = {"w1": N * E, "b1": E, "w2": E, "h": E}
    P = {}
    for k in outs:
        P[k] = torch.randn(outs[k], S, generator=g) * 0.2
        P[k + "_b"] = torch.randn(outs[k], generator=g) * 0.2
    P["b2_w"] = torch.randn(1, E, generator=g)
    P["b2_b"] = torch.randn(1, generator=g)
    s = torch.randn(R, S, generator=g)
    q = torch.randn(R, N, generator=g)
    with torch.no_grad():
        rnd = lambda t: t.bfloat16().float()
        hy = {k: F.linear(rnd(s), rnd(P[k]), P[k + "_b"]) for k in outs}
        hid = F.elu(torch.bmm(q.view(R, 1, N), hy["w1"].abs().view(R, N, E)).view(R, E) + hy["b1"])
        qt = (hid * hy["w2"].abs()).sum(1) + F.linear(F.relu(hy["h"]), P["b2_w"], P["b2_b"]).view(R)
    Wd = {k: cu(v, dev) for k, v in P.items()}
    ld = (S + 3) // 4 * 4
    if remap:
        # the learner's view of a replay sample: (T+1)-slot storage of more episodes than the batch, read in place through
        # an episode map and a slot offset (ops.Rows: row r -> storage row emap[r / T] * (T + 1) + r % T + 1)
        T = 120
        Eb, Es = R // T, R // T + 7
        perm = torch.randperm(Es, generator=g)[:Eb]
        store = torch.zeros(Es, T + 1, ld)
        store[:, :, :S] = torch.randn(Es, T + 1, S, generator=g)          # other slots / episodes hold different data
        store[perm, 1:, :S] = s.view(Eb, T, S)
        sd = cu(store.view(Es * (T + 1), ld), dev)
        xs = ops.src(ops.Rows(sd[:, :S], (T, T + 1, 1), cu(perm, dev, torch.int32)))
    else:
        sd = torch.zeros(R, ld, device=dev)
        sd[:, :S] = cu(s, dev)
        xs = ops.src(sd[:, :S])
    qd = cu(q, dev)
    res = {}
    from marl_amd import experiments
    # res16: 16-row tiles (the default); res32: 32-row tiles, transposed product, mixing in registers (opt-in); stream: the streaming kernel
    for mode, sw in (("res32", dict(wide_res=1, wide_res32=1)), ("res16", dict(wide_res=1, wide_res32=0)), ("stream", dict(wide_res=0, wide_res32=0))):
        with experiments.override(**sw):
            out = torch.full((R,), 9.0, device=dev)
            ops.qmix_wide_fwd(ops.qmix_weights(Wd), xs, qd, out, R, N, S, E, bf16=True)
            res[mode] = out.cpu()
    scale = max(1.0, float(qt.abs().max()))
    for mode in ("res32", "res16"):
        close(res[mode], qt, 1e-4 * scale, 1e-4, msg="q_tot (%s)" % mode)
        close(res[mode], res["stream"], 2e-6 * scale, 1e-5, msg="%s vs streaming kernel" % mode)


@pytest.mark.parametrize("R,N,S,bf16", [(333, 10, 322, False), (64, 10, 322, False), (5000, 10, 322, False), (100, 3, 50, False),
                                         (1000, 5, 120, False), (333, 10, 322, True), (5000, 10, 322, True), (130, 4, 352, True), (300, 4, 384, False),
                                         (333, 10, 322, "fwd"), (5000, 10, 322, "fwd")])
def test_qmix_wide(dev, R, N, S, bf16):
    """wide-state fused QMIX (csrc/qmix_wide.hip: streamed hypernet weights, d(hypernet output) + tall-skinny weight
    gradient GEMM) vs torch-CPU autograd.  fp32: 1e-4.  bf16: the reference is torch-CPU with the hypernet operands
    rounded to bf16 - an EXTERNAL reference for the reduced-precision mode, so the tolerance stays tight (products of
    bf16 values are exact in fp32; only the accumulation order differs); the weight gradient uses the unrounded states
    (straight-through), compared at 2e-2 of its scale.  bf16 = "fwd": bf16 operands in the hypernet GEMM only (flags = 1 of the
    C-ABI, without the weight-gradient bit): the weight-gradient GEMM is fp32 on the unrounded states, held to 1e-4."""
    from marl_amd import ops
    E = 32
    wg32 = bf16 == "fwd"
    bf16 = bool(bf16)
    assert ops.qmix_wide_supported(N, S, E)
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
    qt = _qmix_reference(P, s, q, gq, N, E, bf16, wgrad_fp32=wg32)
    Wd = {k: cu(v.detach(), dev) for k, v in P.items()}
    base = {k: torch.randn(v.shape, generator=g) for k, v in P.items()}      # gradients accumulate
    Gd = {k: cu(v, dev) for k, v in base.items()}
    ld = (S + 3) // 4 * 4
    sd = torch.zeros(R, ld, device=dev)
    sd[:, :S] = cu(s, dev)
    xs = ops.src(sd[:, :S])
    out = torch.full((R,), 9.0, device=dev)
    qd = cu(q.detach(), dev)
    ops.qmix_wide_fwd(ops.qmix_weights(Wd), xs, qd, out, R, N, S, E, bf16=bf16)
    scale_o = max(1.0, float(qt.detach().abs().max()))
    close(out, qt, 1e-4 * scale_o, 1e-4, msg="q_tot")
    dq = torch.full((R, N), 9.0, device=dev)
    ops.qmix_wide_bwd(ops.qmix_weights(Wd), xs, qd, cu(gq, dev), dq, ops.qmix_weights(Gd), R, N, S, E, bf16=bf16, wgrad_bf16=not wg32)
    close(dq, q.grad, 1e-4 * max(1.0, float(q.grad.abs().max())), 1e-4, msg="dq")
    # |.| of the hypernet outputs w1 / w2 and relu of h have kinks at 0: where an output is within fp32 rounding of 0 the two
    # summation orders may pick different one-sided derivatives (this seed has one: row 669, column 120: 6e-8).  Such
    # (row, column) pairs change one row of that segment's weight gradient; those rows are excluded (at most 2 per case).
    with torch.no_grad():
        rnd = (lambda t: t.bfloat16().float()) if bf16 else (lambda t: t)
        kink = {k: ((F.linear(rnd(s), rnd(P[k]), P[k + "_b"]).abs() < 2e-6).any(0)) for k in ("w1", "w2", "h")}
    assert sum(int(v.sum()) for v in kink.values()) <= 2
    for k, v in P.items():
        want = v.grad
        got = (Gd[k] - cu(base[k], dev)).cpu()
        seg = k[:-2] if k.endswith("_b") else k
        if seg in kink and kink[seg].any():
            keep = ~kink[seg]
            want, got = want[keep], got[keep]
        sc = max(1.0, float(want.abs().max()))
        loose = bf16 and not wg32
        tol = (2e-2 if loose and k in outs else 1e-4) * sc
        close(got, want, tol, 2e-2 if loose else 1e-4, msg=k)


@pytest.mark.parametrize("rows,S,N3", [(600, 322, 320), (300, 216, 256), (200, 120, 160)])
def test_mlp3_wide_head_in_column_blocks(dev, rows, S, N3):
    """hyper_w1 of QMixMixer(two_hyper_layers) on MMM2 / 3s5z-sized maps: state -> 64 -> n_agents * 32 outputs evaluated as
    column blocks of <= 160 that SHARE layer 1 (element stride 0 between the groups): outputs and the three gradients vs torch."""
    from marl_amd import ops
    g = torch.Generator().manual_seed(rows + N3)
    ld = (S + 3) // 4 * 4
    x = torch.randn(rows, S, generator=g)
    xd = cu(torch.cat([x, torch.zeros(rows, ld - S)], 1), dev)[:, :S]
    l0, l2 = torch.nn.Linear(S, 64), torch.nn.Linear(64, N3)
    with torch.no_grad():
        for p in list(l0.parameters()) + list(l2.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    d0, d2 = torch.nn.Linear(S, 64).to(dev), torch.nn.Linear(64, N3).to(dev)
    d0.load_state_dict(l0.state_dict()); d2.load_state_dict(l2.state_dict())
    for p in list(d0.parameters()) + list(d2.parameters()):
        p.requires_grad_(False)
        p.grad = torch.zeros_like(p)
    xs = ops.src(xd)
    w, G, n3g = ops.mlp3_wide_head(d0, d2)
    assert G == (N3 + 159) // 160 and G * n3g == N3 and ops.mlp3_supported(xs, S, 64, 0, n3g, G)
    wid = N3 + 96
    hy = torch.full((rows, wid), 7.0, device=dev)
    hs = torch.empty(ops.mlp3_save_floats(rows, False, G), device=dev)
    ops.mlp3_fwd(w, xs, hy[:, 32:32 + N3], rows, S, n3g, G, hsave=hs)
    assert bool((hy[:, :32] == 7).all()) and bool((hy[:, 32 + N3:] == 7).all())
    y = l2(torch.relu(l0(x)))
    close(hy[:, 32:32 + N3], y, 2e-4)
    dY = torch.randn(rows, N3, generator=g)
    dhy = torch.zeros(rows, wid, device=dev)
    dhy[:, 32:32 + N3] = cu(dY, dev)
    ops.mlp3_bwd(w, xs, dhy[:, 32:32 + N3], ops.mlp3_wide_head(d0, d2, grad=True)[0], rows, S, n3g, G, hsave=hs)
    y.backward(dY)
    for name, got, ref in (("W1", d0.weight.grad, l0.weight.grad), ("b1", d0.bias.grad, l0.bias.grad),
                           ("W3", d2.weight.grad, l2.weight.grad), ("b3", d2.bias.grad, l2.bias.grad)):
        scale = max(1.0, float(ref.abs().max()))
        close(got / scale, ref / scale, 3e-4, 1e-4, msg=name)
