"""Mixing networks (mirror of reference network/mixer.py) on the HIP kernels.

Each class keeps the reference's constructor signature, parameter names (state_dict keys) and
``forward`` signature; ``hip_forward`` / ``hip_backward`` are what the learners call: explicit
forward with saved activations, then an explicit backward that accumulates into ``p.grad``
(views of the learner's flat gradient buffer).  No torch autograd anywhere.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..hostutil import Lin, lin_of, require_cuda, to_dev, onehot_to_index


class _Scratch:
    """Cached device buffers keyed by (name, shape)."""

    def __init__(self):
        self.d = {}

    def get(self, name, shape, device, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.d.get(key)
        if t is None or t.device != device:
            t = torch.empty(*shape, dtype=dtype, device=device)
            self.d[key] = t
        return t

    def get_flat(self, name, n, device):
        """>= n floats, grow-only and keyed by NAME alone: the kept-activation buffers of the fused head families are gigabytes
        at the large batches, and max_episode_len - hence `rows` - may differ from update to update (one buffer per distinct
        shape would pile them up)."""
        key = (name, "flat")
        t = self.d.get(key)
        if t is None or t.device != device or t.numel() < n:
            if t is not None:
                # a captured hipGraph (algorithm/common.py:GraphedUpdate) has the old buffer's address baked in: moving the
                # shared workspace generation makes it drop the graph and capture again instead of writing into freed storage
                ops.WS.gen += 1
            t = torch.empty(max(int(n), 1), dtype=torch.float32, device=device)
            self.d[key] = t
        return t

    def get_rows(self, name, rows, width, device):
        """(rows, width) fp32 view whose row stride is rounded up to 4 floats: every row starts on a 16-byte boundary,
        so the GEMM kernels take their vector-load paths (QTRAN's 78-wide intermediates)."""
        ld = (width + 3) // 4 * 4
        return self.get(name, (rows, ld), device)[:, :width]


class _Precision:
    """Operand precision of a mixer's GEMMs, read from ITS args on every call (``args.mixer_dtype``: "fp32" exact -
    default - or "bf16" operands with fp32 accumulation on the bf16 matrix cores, BASELINE config 5).  Per object:
    two learners with different settings can live in one process."""

    def _bf16(self):
        return getattr(self.args, "mixer_dtype", "fp32") == "bf16"

    def _x6mode(self):
        """args.gemm_mode = "bf16x6": the mixer's dense products as bf16x6 splits where a split kernel exists"""
        return getattr(self.args, "gemm_mode", DEFAULT_GEMM_MODE) == "bf16x6"

    def _wgrad_bf16(self):
        """with mixer_dtype "bf16" the wide-state mixer's weight-gradient GEMM takes bf16 operands too (its own flag bit of
        the C-ABI) unless args.mixer_wgrad_dtype = "fp32" keeps that one GEMM on v_mfma_f32_16x16x4_f32"""
        return self._bf16() and getattr(self.args, "mixer_wgrad_dtype", "bf16") != "fp32"

    def _lin(self, module):
        return lin_of(module, self._bf16())


def _mlp(dims, sizes):
    """nn.Sequential(Linear, ReLU, Linear, ...) with the reference's index names 0,2,4."""
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers) if len(layers) > 1 else layers[0]


def _linears(seq):
    return [m for m in (seq if isinstance(seq, nn.Sequential) else [seq]) if isinstance(m, nn.Linear)]


# =====================================================================================
class VDNMixer(_Precision, nn.Module):
    """reference network/mixer.py:9-16."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self._s = _Scratch()

    def hip_forward(self, q, s, rows, ctx=None, tag="e"):
        out = self._s.get("qtot" + tag, (rows,), q.device)
        ops.agent_sum(q, out, rows, self.args.n_agents, 1)
        return out

    def hip_backward(self, ctx, dq_tot, rows):
        dq = self._s.get("dq", (rows, self.args.n_agents), dq_tot.device)
        ops.agent_bcast(dq_tot, dq, rows, self.args.n_agents, 1)
        return dq

    def forward(self, q_values, states=None):
        dev = require_cuda("VDNMixer")
        B = q_values.shape[0]
        q = to_dev(q_values, dev).reshape(-1, self.args.n_agents)
        return self.hip_forward(q, None, q.shape[0]).clone().view(B, -1, 1)


# =====================================================================================
class QMixMixer(_Precision, nn.Module):
    """reference network/mixer.py:21-80 (hypernetwork-generated monotonic mixer)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        S, N, E, HH = args.state_shape, args.n_agents, args.qmix_hidden_dim, args.hyper_hidden_dim
        if args.two_hyper_layers:
            self.hyper_w1 = _mlp(None, [S, HH, N * E])
            self.hyper_w2 = _mlp(None, [S, HH, E])
        else:
            self.hyper_w1 = nn.Linear(S, N * E)
            self.hyper_w2 = nn.Linear(S, E)
        self.hyper_b1 = nn.Linear(S, E)
        self.hyper_b2 = _mlp(None, [S, E, 1])
        self._s = _Scratch()

    def _fused_ok(self, xs):
        a = self.args
        return (not a.two_hyper_layers and not getattr(self, "no_fused", False) and not self._bf16()
                and ops.qmix_fused_supported(a.n_agents, a.state_shape, a.qmix_hidden_dim)
                and xs.ld0 % 4 == 0 and (xs.p0 or 0) % 16 == 0 and not (xs.k1 or xs.nhot or xs.nid or xs.m0))

    def _wide_ok(self, xs):
        """wide-state path (csrc/qmix_wide.hip): weights streamed from L2, fp32 or bf16 operands (args.mixer_dtype)"""
        a = self.args
        S = a.state_shape
        return (not a.two_hyper_layers and not getattr(self, "no_fused", False)
                and ops.qmix_wide_supported(a.n_agents, S, a.qmix_hidden_dim)
                and xs.ld0 % 4 == 0 and xs.ld0 >= (S + 3) // 4 * 4 and (xs.p0 or 0) % 16 == 0
                and not (xs.k1 or xs.nhot or xs.nid or xs.m0))

    def _fused_hyper(self, seq, xs, grad=False):
        """hyper_w1 / hyper_w2 of the two_hyper_layers mixer (state -> hyper_hidden_dim -> N*E | E, mixer.py:36-43) on the fused
        head kernels (csrc/mlp3_fused.hip, wide outputs): (weights, groups, outputs per group) or None."""
        if self._bf16() or getattr(self, "no_fused", False) or not _keep_hidden():
            return None
        l0, l2 = _linears(seq)
        wh = ops.mlp3_wide_head(l0, l2, grad=grad)
        if wh is None or not ops.mlp3_supported(xs, l0.in_features, l0.out_features, 0, wh[2], wh[1]):
            return None
        return wh

    def _fused_tensors(self, grad=False):
        b20, b22 = _linears(self.hyper_b2)
        pick = (lambda p: p.grad) if grad else (lambda p: p.data)
        return {"w1": pick(self.hyper_w1.weight), "w1_b": pick(self.hyper_w1.bias),
                "b1": pick(self.hyper_b1.weight), "b1_b": pick(self.hyper_b1.bias),
                "w2": pick(self.hyper_w2.weight), "w2_b": pick(self.hyper_w2.bias),
                "h": pick(b20.weight), "h_b": pick(b20.bias), "b2_w": pick(b22.weight), "b2_b": pick(b22.bias)}

    def _fused_struct(self, grad=False):
        """marl_qmix_weights_t of the parameters (or their gradients), rebuilt only when a tensor's storage moved"""
        t = self._fused_tensors(grad)
        key = tuple(v.data_ptr() for v in t.values())
        cache = self.__dict__.setdefault("_fs_cache", {})
        c = cache.get(grad)
        if c is None or c[0] != key:
            c = cache[grad] = (key, ops.qmix_weights(t))
        return c[1]

    def __deepcopy__(self, memo):
        # the cached ctypes structs point at THIS module's storage: a copy (target mixer) starts without them
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy as _copy
        for k, v in self.__dict__.items():
            if k != "_fs_cache":
                new.__dict__[k] = _copy.deepcopy(v, memo)
        return new

    def hip_forward(self, q, s, rows, ctx=None, tag="e"):
        a = self.args
        N, E, HH = a.n_agents, a.qmix_hidden_dim, a.hyper_hidden_dim
        dev = q.device
        xs = ops.src(s)
        if self._fused_ok(xs):
            # one kernel: hypernet GEMMs (weights in registers) + mixing; nothing 256-wide touches HBM
            qtot = self._s.get("qtot" + tag, (rows,), dev)
            ops.qmix_fused_fwd(self._fused_struct(), xs, q, qtot, rows, N, a.state_shape, E, x6=self._x6mode())
            if ctx is not None:
                ctx.update(q=q, s=s, fused=True)
            return qtot
        if self._wide_ok(xs):
            # wide states (MMM2): same fusion with the 416 x 322 hypernet streamed from L2; bf16 operands on request
            qtot = self._s.get("qtot" + tag, (rows,), dev)
            ops.qmix_wide_fwd(self._fused_struct(), xs, q, qtot, rows, N, a.state_shape, E, bf16=self._bf16())
            if ctx is not None:
                ctx.update(q=q, s=s, wide=True)
            return qtot
        wid = N * E + 3 * E
        hy = self._s.get("hy" + tag, (rows, wid), dev)
        b2 = self._s.get("b2" + tag, (rows, 1), dev)
        qtot = self._s.get("qtot" + tag, (rows,), dev)
        hw1 = hw2 = None
        if a.two_hyper_layers:
            kept, hid = {}, {}
            for name, seq, cols in (("w1", self.hyper_w1, slice(0, N * E)), ("w2", self.hyper_w2, slice(N * E + E, N * E + 2 * E))):
                wh = self._fused_hyper(seq, xs)
                if wh is not None:
                    # both layers in one launch; the hidden activations go to HBM only as the backward's fragments
                    w, G, n3g = wh
                    hs = None
                    if ctx is not None:
                        hs = kept[name] = self._s.get_flat("hs_" + name, ops.mlp3_save_floats(rows, False, G), dev)
                    ops.mlp3_fwd(w, xs, hy[:, cols], rows, a.state_shape, n3g, G, hsave=hs)
                    continue
                l0, l2 = _linears(seq)
                hbuf = hid[name] = self._s.get("h" + name + tag, (rows, HH), dev)
                self._lin(l0).fwd(xs, hbuf, rows, act=1)
                self._lin(l2).fwd(ops.src(hbuf), hy[:, cols], rows)
            hw1, hw2 = hid.get("w1"), hid.get("w2")
        else:
            self._lin(self.hyper_w1).fwd(xs, hy[:, :N * E], rows)
            self._lin(self.hyper_w2).fwd(xs, hy[:, N * E + E:N * E + 2 * E], rows)
        b20, b22 = _linears(self.hyper_b2)
        tail = self._tail_ok(s)
        if tail:
            # hyper_b1 and hyper_b2.0 in one pass over s, straight into hy; hyper_b2.2 inside the mixing kernel: no marl_linear
            ops.qmix_tail_fwd(s, rows, a.state_shape, self.hyper_b1.weight.data, self.hyper_b1.bias.data, b20.weight.data,
                              b20.bias.data, hy, N * E, N * E + 2 * E)
            ops.qmix_mix_fwd(hy, None, q, qtot, rows, N, E, w22=b22.weight.data, b22=b22.bias.data)
        else:
            self._lin(self.hyper_b1).fwd(xs, hy[:, N * E:N * E + E], rows)
            self._lin(b20).fwd(xs, hy[:, N * E + 2 * E:], rows, act=1)
            self._lin(b22).fwd(ops.src(hy[:, N * E + 2 * E:]), b2, rows)
            ops.qmix_mix_fwd(hy, b2, q, qtot, rows, N, E)
        if ctx is not None:
            ctx.update(hy=hy, q=q, s=s, hw1=hw1, hw2=hw2, kept=kept if a.two_hyper_layers else {}, tail=tail)
        return qtot

    def _tail_ok(self, s):
        """the bias layers hyper_b1 / hyper_b2 on the row kernel (csrc/qtran_fused.hip: marl_qmix_tail_fwd) - fp32 only"""
        a = self.args
        return (not self._bf16() and not getattr(self, "no_fused", False)
                and ops.qmix_tail_supported(a.state_shape, a.qmix_hidden_dim, s)
                and self.hyper_b1.weight.data.data_ptr() % 4 == 0)

    def loss_backward_fused(self, s):
        """True when hip_loss_backward covers this shape (one of the two fused kernel families)."""
        xs = ops.src(s)
        return self._fused_ok(xs) or self._wide_ok(xs)

    def hip_loss_backward(self, q, s, rows, q_tot_tgt, r, term, padded, gamma, loss2, q_tot=None):
        """Forward + TD loss + backward of the mixer in ONE launch (csrc/qmix_fused.hip, LOSS variant): the backward pass
        recomputes q_tot anyway, so the eval mixer's forward launch, the loss launch and its reduction are not needed
        (q_learner.py:112-127).  loss2 (2 floats, accumulated into): sum (mask td)^2, sum mask.  Returns dL/dq (rows, N);
        hypernet gradients are accumulated into .grad; q_tot (rows) is written when given."""
        a = self.args
        N, E = a.n_agents, a.qmix_hidden_dim
        dq = self._s.get("dq", (rows, N), q.device)
        xs = ops.src(s)
        if self._fused_ok(xs):
            ops.qmix_fused_loss_bwd(self._fused_struct(), xs, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq,
                                    self._fused_struct(grad=True), loss2, rows, N, a.state_shape, E, x6=self._x6mode())
        else:
            ops.qmix_wide_loss_bwd(self._fused_struct(), xs, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq,
                                   self._fused_struct(grad=True), loss2, rows, N, a.state_shape, E, bf16=self._bf16(),
                                   wgrad_bf16=self._wgrad_bf16())
        return dq

    def hip_backward(self, ctx, dq_tot, rows):
        a = self.args
        N, E, HH = a.n_agents, a.qmix_hidden_dim, a.hyper_hidden_dim
        if ctx.get("fused"):
            q, s = ctx["q"], ctx["s"]
            dq = self._s.get("dq", (rows, N), q.device)
            ops.qmix_fused_bwd(self._fused_struct(), ops.src(s), q, dq_tot, dq,
                               self._fused_struct(grad=True), rows, N, a.state_shape, E, x6=self._x6mode())
            return dq
        if ctx.get("wide"):
            q, s = ctx["q"], ctx["s"]
            dq = self._s.get("dq", (rows, N), q.device)
            ops.qmix_wide_bwd(self._fused_struct(), ops.src(s), q, dq_tot, dq, self._fused_struct(grad=True), rows, N,
                              a.state_shape, E, bf16=self._bf16(), wgrad_bf16=self._wgrad_bf16())
            return dq
        hy, q, s = ctx["hy"], ctx["q"], ctx["s"]
        dev = q.device
        dhy = self._s.get("dhy", hy.shape, dev)
        db2 = self._s.get("db2", (rows, 1), dev)
        dq = self._s.get("dq", (rows, N), dev)
        b20, b22 = _linears(self.hyper_b2)
        ops.qmix_mix_bwd(hy, q, dq_tot, dhy, db2, dq, rows, N, E, w22=b22.weight.data if ctx.get("tail") else None)
        xs = ops.src(s)
        hb, dhb = hy[:, N * E + 2 * E:], dhy[:, N * E + 2 * E:]
        self._lin(b22).wgrad(db2, ops.src(hb), rows)
        if not ctx.get("tail"):
            self._lin(b22).bwd_x(db2, dhb, rows)
        self._lin(b20).wgrad(dhb, xs, rows, Yact=hb)
        self._lin(self.hyper_b1).wgrad(dhy[:, N * E:N * E + E], xs, rows)
        if a.two_hyper_layers:
            for name, seq, hbuf, cols in (("w1", self.hyper_w1, ctx["hw1"], slice(0, N * E)),
                                          ("w2", self.hyper_w2, ctx["hw2"], slice(N * E + E, N * E + 2 * E))):
                hs = ctx["kept"].get(name)
                if hs is not None:
                    wh, gh = self._fused_hyper(seq, xs), self._fused_hyper(seq, xs, grad=True)
                    if wh is None or gh is None:
                        # the forward kept only the fused kernels' fragments: there is nothing to fall back to
                        raise RuntimeError("QMixMixer: the fused hypernet head %r kept its activations in the forward but its "
                                           "backward cannot be launched (.grad tensors missing / not contiguous / not 16-byte "
                                           "aligned, or MARL_MLP3_KEEP changed between forward and backward)" % name)
                    w, G, n3g = wh
                    ops.mlp3_bwd(w, xs, dhy[:, cols], gh[0], rows, a.state_shape, n3g, G, hsave=hs)
                    continue
                l0, l2 = _linears(seq)
                self._lin(l2).wgrad(dhy[:, cols], ops.src(hbuf), rows)
                dh = self._s.get("dhw", (rows, HH), dev)
                self._lin(l2).bwd_x(dhy[:, cols], dh, rows)
                self._lin(l0).wgrad(dh, xs, rows, Yact=hbuf)
        else:
            self._lin(self.hyper_w1).wgrad(dhy[:, :N * E], xs, rows)
            self._lin(self.hyper_w2).wgrad(dhy[:, N * E + E:N * E + 2 * E], xs, rows)
        return dq

    def forward(self, q_values, states):
        dev = require_cuda("QMixMixer")
        self.to(dev)
        B = q_values.shape[0]
        q = to_dev(q_values, dev).reshape(-1, self.args.n_agents)
        s = to_dev(states, dev).reshape(-1, self.args.state_shape)
        return self.hip_forward(q, s, q.shape[0]).clone().view(B, -1, 1)


# =====================================================================================
class DMAQ_SI_Weight(nn.Module):
    """lambda-net of QPLEX (reference network/mixer.py:85-171): num_kernel heads x
    {key, agents, action} extractors."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.n_agents, self.n_actions = args.n_agents, args.n_actions
        self.state_dim = int(np.prod(args.state_shape))
        self.action_dim = args.n_agents * args.n_actions
        self.state_action_dim = self.state_dim + self.action_dim
        self.num_kernel = args.num_kernel
        nl = getattr(args, "adv_hypernet_layers", 1)
        if nl not in (1, 2, 3):
            raise Exception("Error setting number of adv hypernet layers.")
        AE = args.adv_hypernet_embed
        hid = [AE] * (nl - 1)
        self.key_extractors = nn.ModuleList()
        self.agents_extractors = nn.ModuleList()
        self.action_extractors = nn.ModuleList()
        for _ in range(self.num_kernel):
            self.key_extractors.append(_mlp(None, [self.state_dim] + hid + [1]))
            self.agents_extractors.append(_mlp(None, [self.state_dim] + hid + [self.n_agents]))
            self.action_extractors.append(_mlp(None, [self.state_action_dim] + hid + [self.n_agents]))

    def families(self):
        return (("key", self.key_extractors, 1), ("ag", self.agents_extractors, self.n_agents),
                ("ac", self.action_extractors, self.n_agents))

    def forward(self, states, actions):
        raise RuntimeError("DMAQ_SI_Weight is evaluated inside DMAQer.hip_forward (fused lambda-net path)")


# args.gemm_mode when the caller's args do not carry one: "f32" (v_mfma_f32_16x16x4_f32 everywhere).  "bf16x6" is opt-in; the GPU test
# suite sets this attribute to run every QPLEX parity case on the split kernels (tests/conftest.py, MARL_TEST_GEMM_MODE).
DEFAULT_GEMM_MODE = "f32"


def _keep_hidden():
    """the fused head families keep their hidden activations for the backward (MARL_MLP3_KEEP=0: recompute them there)"""
    from .. import experiments
    return experiments.get("mlp3_keep") != 0


def _head_stride(mods, attr):
    """element stride between consecutive heads' tensors if uniform (flat parameter buffer), else None."""
    ts = [getattr(m, attr) for m in mods]
    if len(ts) == 1:
        return 0
    d = [(ts[i + 1].data_ptr() - ts[i].data_ptr()) for i in range(len(ts) - 1)]
    if any(x != d[0] for x in d) or d[0] % 4 != 0 or d[0] <= 0:
        return None
    return d[0] // 4


class DMAQer(_Precision, nn.Module):
    """QPLEX duplex dueling mixer (reference network/mixer.py:173-288)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.n_agents, self.n_actions = args.n_agents, args.n_actions
        self.state_dim = int(np.prod(args.state_shape))
        self.action_dim = args.n_agents * args.n_actions
        self.state_action_dim = self.state_dim + self.action_dim + 1
        self.embed_dim = args.mixing_embed_dim
        HE = args.hypernet_embed
        self.hyper_w_final = _mlp(None, [self.state_dim, HE, self.n_agents])
        self.V = _mlp(None, [self.state_dim, HE, self.n_agents])
        self.si_weight = DMAQ_SI_Weight(args)
        self._s = _Scratch()

    # ---- grouped dense layer over the K heads of one extractor family
    def _layer(self, mods, li, x, x_gs, Y, y_gs, rows, act):
        lins = [_linears(m)[li] for m in mods]
        K = len(lins)
        gw, gb = _head_stride(lins, "weight"), _head_stride(lins, "bias")
        N_, K_ = lins[0].weight.shape
        if gw is not None and gb is not None:
            grp = ops.group(K, x0=x_gs, w=gw, b=gb, y=y_gs)
            ops.linear(x, lins[0].weight.data, lins[0].bias.data, Y, rows, N_, K_, act=act, grp=grp, bf16=self._bf16())
        else:   # parameters not in one flat buffer: one launch per head
            for k, l in enumerate(lins):
                xk = x if x_gs == 0 else ops.src(self._xview(x, k, x_gs, K_))
                ops.linear(xk, l.weight.data, l.bias.data, Y[:, k * y_gs:(k + 1) * y_gs], rows, N_, K_, act=act, bf16=self._bf16())

    @staticmethod
    def _xview(x, k, gs, width):
        t = x._keep[0]
        return t[:, k * gs:k * gs + width]

    def _fused_family(self, mods, x_in, nout, grad=False):
        """marl_mlp3_weights_t of an extractor family when the fused three-layer kernel covers it, else None."""
        if self._bf16() or getattr(self, "no_fused", False):
            return None
        heads = [_linears(m) for m in mods]
        if len(heads[0]) != 3:
            return None
        l0, l1, l2 = heads[0]
        if not ops.mlp3_supported(x_in, l0.in_features, l0.out_features, l1.out_features, nout, len(heads)):
            return None
        if ops.mlp3_needs_kept(x_in, l0.in_features) and not _keep_hidden():
            return None
        return ops.mlp3_weights(heads, grad=grad)

    def _x6(self, x_in, K1, nout, groups):
        """the family runs on the bf16x6 split kernels (csrc/mlp3_x6.hip): opt-in args.gemm_mode = "bf16x6" - fp32-accurate
        products on the bf16 matrix cores; the default "f32" keeps v_mfma_f32_16x16x4_f32"""
        return (getattr(self.args, "gemm_mode", DEFAULT_GEMM_MODE) == "bf16x6" and _keep_hidden()
                and ops.mlp3_x6_supported(x_in, K1, 64, 64, nout, groups))

    def _kept(self, name, rows, three, groups, dev):
        """buffer for the hidden activations a fused family keeps for its backward (MARL_MLP3_KEEP=0: recompute)."""
        if not _keep_hidden():
            return None
        return self._s.get_flat(name + "_hs", ops.mlp3_save_floats(rows, three, groups), dev)

    def _fused_transform(self, xs, grad=False):
        """hyper_w_final and V (Linear-ReLU-Linear, same shapes) as two heads of the fused kernel, or None."""
        if self._bf16() or getattr(self, "no_fused", False):
            return None
        heads = [_linears(self.hyper_w_final), _linears(self.V)]
        if len(heads[0]) != 2 or not ops.mlp3_supported(xs, self.state_dim, heads[0][0].out_features, 0, self.n_agents, 2):
            return None
        if ops.mlp3_needs_kept(xs, self.state_dim) and not _keep_hidden():
            return None
        return ops.mlp3_weights(heads, grad=grad)

    def _lambda_fwd(self, s, u_idx, rows, tag, keep):
        """raw head outputs key (rows,K), ag (rows,K,N), ac (rows,K,N) + hidden activations."""
        a = self.args
        K, AE, N, A = a.num_kernel, a.adv_hypernet_embed, a.n_agents, a.n_actions
        dev = s.device
        outs = {}
        xs = ops.src(s)
        xsa = ops.src(s, idx=u_idx.view(rows, N), nhot=N, hot_w=A)
        for name, mods, nout in self.si_weight.families():
            x_in = xsa if name == "ac" else xs
            out = self._s.get("%s_out%s" % (name, tag), (rows, K * nout), dev)
            outs[name] = out
            fw = self._fused_family(mods, x_in, nout)
            if fw is not None:
                # one kernel per family: the 10 heads' hidden activations never travel between layers.  When a backward
                # follows they are kept as the backward's MFMA fragments (streamed out once, read once) - recomputing
                # layers 1-2 there was a third of its time
                hs = self._kept(name, rows, True, K, dev) if keep is not None else None
                x6 = self._x6(x_in, ops.src_width(x_in), nout, K)
                ops.mlp3_fwd(fw, x_in, out, rows, ops.src_width(x_in), nout, K, hsave=hs, x6=x6)
                if keep is not None:
                    keep[name + "_h"] = None
                    keep[name + "_hs"] = hs
                    keep[name + "_x6"] = x6
                continue
            nl = len(_linears(mods[0]))
            hs = []
            cur, cur_gs = x_in, 0
            for li in range(nl - 1):
                hbuf = self._s.get("%s_h%d%s" % (name, li, tag), (rows, K * AE), dev)
                self._layer(mods, li, cur, cur_gs, hbuf, AE, rows, act=1)
                hs.append(hbuf)
                cur, cur_gs = ops.src(hbuf, k0=AE), AE
            self._layer(mods, nl - 1, cur, cur_gs, out, nout, rows, act=0)
            if keep is not None:
                keep[name + "_h"] = hs
        return outs

    def hip_forward(self, q, s, rows, u_idx=None, max_q=None, ctx=None, tag="e"):
        """q (rows,N) chosen Qs; returns (v_tot, a_tot) - a_tot None when max_q is None (is_v only).
        u_idx (rows*N) int32 actions whose one-hot feeds the action extractors."""
        a = self.args
        N, K, HE = a.n_agents, a.num_kernel, a.hypernet_embed
        dev = q.device
        xs = ops.src(s)
        w0, w2 = _linears(self.hyper_w_final)
        v0, v2 = _linears(self.V)
        hw = hv = None
        tw = self._fused_transform(xs)
        if tw is not None:
            # hyper_w_final and V as ONE two-head launch of the fused kernel; wv[0] = w_raw, wv[1] = v
            wv = self._s.get("wv" + tag, (2, rows, N), dev)
            wv_hs = self._kept("wv", rows, False, 2, dev) if ctx is not None else None
            ops.mlp3_fwd(tw, xs, wv, rows, self.state_dim, N, 2, hsave=wv_hs)
            if ctx is not None:
                ctx["wv_hs"] = wv_hs
            w_raw, v = wv[0], wv[1]
        else:
            hw = self._s.get("hw" + tag, (rows, HE), dev)
            hv = self._s.get("hv" + tag, (rows, HE), dev)
            w_raw = self._s.get("wraw" + tag, (rows, N), dev)
            v = self._s.get("v" + tag, (rows, N), dev)
            self._lin(w0).fwd(xs, hw, rows, act=1)
            self._lin(w2).fwd(ops.src(hw), w_raw, rows)
            self._lin(v0).fwd(xs, hv, rows, act=1)
            self._lin(v2).fwd(ops.src(hv), v, rows)
        v_tot = self._s.get("vtot" + tag, (rows,), dev)
        a_tot = lam = None
        heads = {}
        if max_q is not None:
            heads = self._lambda_fwd(s, u_idx, rows, tag, ctx)
            a_tot = self._s.get("atot" + tag, (rows,), dev)
            lam = self._s.get("lam" + tag, (rows, N), dev)
        ops.qplex_mix_fwd(w_raw, v, q, max_q, heads.get("key"), heads.get("ag"), heads.get("ac"), v_tot, a_tot, lam,
                          rows, N, K, a.weighted_head, a.is_minus_one)
        if ctx is not None:
            ctx.update(q=q, s=s, u_idx=u_idx, max_q=max_q, hw=hw, hv=hv, w_raw=w_raw, v=v, heads=heads, lam=lam)
        return v_tot, a_tot

    def hip_backward(self, ctx, g, rows):
        """g = dL/d(v_tot + a_tot) (rows).  Returns dq (rows,N); accumulates parameter grads."""
        a = self.args
        N, K, AE, A, HE = a.n_agents, a.num_kernel, a.adv_hypernet_embed, a.n_actions, a.hypernet_embed
        s, q = ctx["s"], ctx["q"]
        dev = q.device
        heads = ctx["heads"]
        dq = self._s.get("dq", (rows, N), dev)
        dwv = self._s.get("dwv", (2, rows, N), dev)
        dw_raw, dv = dwv[0], dwv[1]
        douts = {"key": self._s.get("dkey", (rows, K), dev), "ag": self._s.get("dag", (rows, K * N), dev),
                 "ac": self._s.get("dac", (rows, K * N), dev)}
        ops.qplex_mix_bwd(ctx["w_raw"], q, ctx["max_q"], heads["key"], heads["ag"], heads["ac"], g, dq, dw_raw, dv,
                          douts["key"], douts["ag"], douts["ac"], rows, N, K, a.weighted_head, a.is_minus_one)
        xs = ops.src(s)
        # transformation nets
        if ctx["hw"] is None:
            ops.mlp3_bwd(self._fused_transform(xs), xs, dwv, self._fused_transform(xs, grad=True), rows, self.state_dim, N, 2,
                         hsave=ctx.get("wv_hs"))
        else:
            for seq, hbuf, dout in ((self.hyper_w_final, ctx["hw"], dw_raw), (self.V, ctx["hv"], dv)):
                l0, l2 = _linears(seq)
                self._lin(l2).wgrad(dout, ops.src(hbuf), rows)
                dh = self._s.get("dh_t", (rows, HE), dev)
                self._lin(l2).bwd_x(dout, dh, rows)
                self._lin(l0).wgrad(dh, xs, rows, Yact=hbuf)
        # lambda-net, family by family, heads batched
        xsa = ops.src(s, idx=ctx["u_idx"].view(rows, N), nhot=N, hot_w=A)
        for name, mods, nout in self.si_weight.families():
            x_in = xsa if name == "ac" else xs
            hs = ctx[name + "_h"]
            if hs is None:      # fused forward: the backward recomputes the hidden activations on chip
                ops.mlp3_bwd(self._fused_family(mods, x_in, nout), x_in, douts[name],
                             self._fused_family(mods, x_in, nout, grad=True), rows, ops.src_width(x_in), nout, K,
                             hsave=ctx.get(name + "_hs"), x6=bool(ctx.get(name + "_x6")))
                continue
            nl = len(_linears(mods[0]))
            dcur, dcur_gs, gate = douts[name], nout, None
            for li in range(nl - 1, -1, -1):
                lins = [_linears(m)[li] for m in mods]
                gw, gb = _head_stride(lins, "weight"), _head_stride(lins, "bias")
                N_, K_ = lins[0].weight.shape
                if li > 0:
                    xin, xin_gs = ops.src(hs[li - 1], k0=K_), AE
                else:
                    xin, xin_gs = x_in, 0
                assert gw is not None and gb is not None, "QPLEX heads must live in one flat parameter buffer"
                ggw = _head_stride(lins, "weight")
                grp = ops.group(K, x0=xin_gs, w=ggw, b=gb, y=dcur_gs, m0=dcur_gs)
                # gradient buffers follow the parameter layout (views of the learner's flat grad)
                ops.linear_wgrad(dcur, xin, lins[0].weight.grad, lins[0].bias.grad, rows, N_, K_, Yact=gate, grp=grp, bf16=self._bf16())
                if li > 0:
                    dprev = self._s.get("dh_%s%d" % (name, li), (rows, K * AE), dev)
                    gx = ops.group(K, x0=dcur_gs, w=gw, y=AE, m0=dcur_gs)
                    ops.linear(ops.src(dcur, gate=gate, k0=N_), lins[0].weight.data, None, dprev, rows, K_, N_,
                               w_kmajor=True, grp=gx, bf16=self._bf16())
                    dcur, dcur_gs, gate = dprev, AE, hs[li - 1]
        return dq

    def forward(self, agent_qs, states, actions=None, max_q_i=None, is_v=False):
        dev = require_cuda("DMAQer")
        self.to(dev)
        bs = agent_qs.shape[0]
        N = self.n_agents
        q = to_dev(agent_qs, dev).reshape(-1, N)
        s = to_dev(states, dev).reshape(-1, self.state_dim)
        rows = q.shape[0]
        if is_v:
            v_tot, _ = self.hip_forward(q, s, rows, tag="f")
            return v_tot.clone().view(bs, -1, 1)
        u_idx = onehot_to_index(to_dev(actions, dev).reshape(rows, N, self.n_actions)).reshape(-1)
        mq = to_dev(max_q_i, dev).reshape(-1, N)
        ctx = {}
        _, a_tot = self.hip_forward(q, s, rows, u_idx=u_idx, max_q=mq, ctx=ctx, tag="f")
        self.last_lambda = ctx["lam"].clone()          # DMAQ_SI_Weight.forward output (rows, N), reference :155-169
        return a_tot.clone().view(bs, -1, 1)


# =====================================================================================
class _QtranFusedHead:
    """Fused path shared by QtranQBase (one-hot actions, A = n_actions) and QtranV (A = 0): csrc/qtran_fused.hip.
    The subclass provides ``_qt_layers()`` -> (enc.0, enc.2, head.0, head.2, head.4) and ``_qt_actions()``."""

    def _qt_dims(self):
        a = self.args
        A = self._qt_actions()
        AE = a.rnn_hidden_dim + A
        return a.n_agents, A, AE, (AE + 15) // 16 * 16, a.state_shape

    def _qt_ok(self, hidden):
        a = self.args
        N, A, AE, _, _ = self._qt_dims()
        return (not self._bf16() and not getattr(self, "no_fused", False) and a.rnn_hidden_dim == 64
                and a.qtran_hidden_dim == 64 and ops.qtran_supported(N, A, AE) and hidden.is_contiguous()
                and hidden.data_ptr() % 16 == 0)

    def _qt_struct(self):
        ls = self._qt_layers()
        key = tuple(p.data_ptr() for l in ls for p in (l.weight, l.bias))
        c = self.__dict__.get("_qt_cache")
        if c is None or c[0] != key:
            c = self.__dict__["_qt_cache"] = (key, ops.qtran_weights(*ls, self.args.state_shape))
        return c[1]

    def __deepcopy__(self, memo):
        # the cached ctypes struct points at THIS module's storage: a copy (target mixer) starts without it
        import copy as _copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_qt_cache":
                new.__dict__[k] = _copy.deepcopy(v, memo)
        return new

    def state_part(self, s, BT, tag="e", other=None):
        """sp = W_0[:, :S] s + b_0 (BT, 64): the state columns of the head's first layer.  Independent of hidden
        states and actions, so one call serves every evaluation of this network on the same states.  ``other``: a second
        head (the V network beside the joint-Q network) evaluated on the same rows in the same pass over s; returns
        (sp, sp_other) then."""
        q0 = self._qt_layers()[2]
        S = self.args.state_shape
        sp = self._s.get("sp" + tag, (BT, q0.out_features), s.device)
        heads = [(self, sp)]
        if other is not None:
            heads.append((other, other._s.get("sp" + tag, (BT, q0.out_features), s.device)))
        if ops.qtran_state_parts_supported(S, s) and not self._bf16():
            sets = []
            for net, out in heads:
                l0 = net._qt_layers()[2]
                sets.append((l0.weight.data, l0.bias.data, out))
            ops.qtran_state_parts(s, BT, S, sets)
        else:
            for net, out in heads:
                l0 = net._qt_layers()[2]
                Lin(l0.weight.data[:, :S], l0.bias, net._bf16()).fwd(ops.src(s), out, BT)
        return sp if other is None else (sp, heads[1][1])

    def _qt_forward(self, s, hidden, u_idx, BT, ctx, tag, sp, u_idx2=None):
        N, A, AE, AEP, S = self._qt_dims()
        dev = hidden.device
        if sp is None:
            sp = self.state_part(s, BT, tag)
        out = self._s.get("out" + tag, (BT,), dev)
        s1 = e2 = y1 = y2 = None
        if ctx is not None:
            s1, e2 = self._s.get("s1" + tag, (BT, AEP), dev), self._s.get("e2" + tag, (BT, AEP), dev)
            y1, y2 = self._s.get("y1" + tag, (BT, 64), dev), self._s.get("y2" + tag, (BT, 64), dev)
        out2 = None
        if u_idx2 is not None:          # a second action set on the same rows: the first encoder product is shared
            out2 = self._s.get("out2" + tag, (BT,), dev)
            ops.qtran_head_fwd2(self._qt_struct(), hidden, u_idx, u_idx2, sp, out, out2, s1, e2, y1, y2, BT, N, A, AE)
        else:
            ops.qtran_head_fwd(self._qt_struct(), hidden, u_idx if A else None, sp, out, s1, e2, y1, y2, BT, N, A, AE)
        if ctx is not None:
            ctx.update(fused=True, s=s, hidden=hidden, u_idx=u_idx, s1=s1, e2=e2, y1=y1, y2=y2)
        return out if u_idx2 is None else (out, out2)

    def _wgrad_split(self, lin, dY, s, e, BT, S):
        """lin.weight.grad += dY^T [s | e], lin.bias.grad += colsum(dY): column blocks [0, S) and [S, S + width(e))"""
        gw = lin.weight.grad
        K = gw.shape[1]
        if self._bf16() or S % 4:
            self._lin(lin).wgrad(dY, ops.src(s, e), BT)
            return
        ops.linear_wgrad(dY, ops.src(s), gw[:, :S], lin.bias.grad, BT, gw.shape[0], S)
        ops.linear_wgrad(dY, ops.src(e), gw[:, S:], None, BT, gw.shape[0], K - S)

    def _qt_backward(self, ctx, d_out, BT, dhidden, accumulate):
        N, A, AE, AEP, S = self._qt_dims()
        dev = d_out.device
        e0, e2l, q0, q2, q4 = self._qt_layers()
        dy1, dy2 = self._s.get("dy1", (BT, 64), dev), self._s.get("dy2", (BT, 64), dev)
        de2 = self._s.get("de2", (BT, AEP), dev)
        # head chain + agent-level pass: dhidden, and the gradients of encoder layer 1 / the bias of layer 2
        ops.qtran_head_bwd(self._qt_struct(), ctx["hidden"], ctx["u_idx"] if A else None, d_out, ctx["y1"], ctx["y2"],
                           dy1, dy2, de2, dhidden, accumulate, e0.weight.grad, e0.bias.grad, e2l.bias.grad, BT, N, A, AE)
        # row-level weight gradients: reductions over BT rows of tensors the kernel above has just written
        s = ctx["s"]
        if ops.qtran_wgrad_rows_supported(S, AE, s) and not self._bf16():
            ops.qtran_wgrad_rows(s, ctx["s1"], ctx["e2"], ctx["y1"], ctx["y2"], d_out, dy1, dy2, de2, q0.weight.grad, q0.bias.grad,
                                 q2.weight.grad, q2.bias.grad, q4.weight.grad, q4.bias.grad, e2l.weight.grad, BT, S, AE)
            return
        self._lin(q4).wgrad(d_out.view(BT, 1), ops.src(ctx["y2"]), BT)
        self._lin(q2).wgrad(dy2, ops.src(ctx["y1"]), BT)
        # first head layer, [s | esum] -> 64: the state columns and the encoder columns as two reductions - the state part
        # (216 columns on 3s5z) runs on the LDS-staged tall kernel, which the virtual concat of two dense segments does not
        self._wgrad_split(q0, dy1, s, ctx["e2"][:, :AE], BT, S)
        Lin(e2l.weight, None, self._bf16()).wgrad(de2[:, :AE], ops.src(ctx["s1"][:, :AE]), BT)


class QtranQBase(_QtranFusedHead, _Precision, nn.Module):
    """QTRAN-base joint action-value network (reference network/mixer.py:355-388)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        ae = args.rnn_hidden_dim + args.n_actions
        self.hidden_action_encoding = _mlp(None, [ae, ae, ae])
        q_in = args.state_shape + args.n_actions + args.rnn_hidden_dim
        self.q = _mlp(None, [q_in, args.qtran_hidden_dim, args.qtran_hidden_dim, 1])
        self._s = _Scratch()

    def _qt_layers(self):
        return tuple(_linears(self.hidden_action_encoding) + _linears(self.q))

    def _qt_actions(self):
        return self.args.n_actions

    def hip_forward(self, s, hidden, u_idx, BT, ctx=None, tag="e", sp=None, u_idx2=None):
        """s (BT,S); hidden (BT*N,H); u_idx (BT*N) int32 -> joint q (BT).  sp: optional result of state_part(s).
        u_idx2: a second set of actions evaluated on the same states and hidden states -> (q, q2); ctx belongs to the first."""
        if self._qt_ok(hidden):
            return self._qt_forward(s, hidden, u_idx, BT, ctx, tag, sp, u_idx2)
        if u_idx2 is not None:
            q1 = self.hip_forward(s, hidden, u_idx, BT, ctx=ctx, tag=tag, sp=sp)
            return q1, self.hip_forward(s, hidden, u_idx2, BT, tag=tag + "2", sp=sp)
        a = self.args
        N, H, A, Q = a.n_agents, a.rnn_hidden_dim, a.n_actions, a.qtran_hidden_dim
        R, ae = BT * N, H + A
        dev = s.device
        e0, e2 = _linears(self.hidden_action_encoding)
        q0, q2, q4 = _linears(self.q)
        x_ha = ops.src(hidden, idx=u_idx.view(R, 1), nhot=1, hot_w=A)
        e1 = self._s.get_rows("e1" + tag, R, ae, dev)
        e2b = self._s.get_rows("e2" + tag, R, ae, dev)
        esum = self._s.get_rows("esum" + tag, BT, ae, dev)
        y1 = self._s.get("y1" + tag, (BT, Q), dev)
        y2 = self._s.get("y2" + tag, (BT, Q), dev)
        out = self._s.get("out" + tag, (BT, 1), dev)
        self._lin(e0).fwd(x_ha, e1, R, act=1)
        self._lin(e2).fwd(ops.src(e1), e2b, R)
        ops.agent_sum(e2b, esum, BT, N, ae)
        x_q = ops.src(s, esum)
        self._lin(q0).fwd(x_q, y1, BT, act=1)
        self._lin(q2).fwd(ops.src(y1), y2, BT, act=1)
        self._lin(q4).fwd(ops.src(y2), out, BT)
        if ctx is not None:
            ctx.update(s=s, hidden=hidden, u_idx=u_idx, e1=e1, esum=esum, y1=y1, y2=y2)
        return out.view(BT)

    def hip_backward(self, ctx, d_out, BT, dhidden, accumulate):
        """d_out (BT). Adds/writes the gradient wrt hidden into dhidden (BT*N,H)."""
        if ctx.get("fused"):
            return self._qt_backward(ctx, d_out, BT, dhidden, accumulate)
        a = self.args
        N, H, A, Q, S = a.n_agents, a.rnn_hidden_dim, a.n_actions, a.qtran_hidden_dim, a.state_shape
        R, ae = BT * N, H + A
        dev = d_out.device
        e0, e2 = _linears(self.hidden_action_encoding)
        q0, q2, q4 = _linears(self.q)
        s, hidden, u_idx, e1, esum, y1, y2 = (ctx[k] for k in ("s", "hidden", "u_idx", "e1", "esum", "y1", "y2"))
        g = d_out.view(BT, 1)
        dy2 = self._s.get("dy2", (BT, Q), dev)
        dy1 = self._s.get("dy1", (BT, Q), dev)
        desum = self._s.get_rows("desum", BT, ae, dev)
        de2 = self._s.get_rows("de2", R, ae, dev)
        de1 = self._s.get_rows("de1", R, ae, dev)
        self._lin(q4).wgrad(g, ops.src(y2), BT)
        self._lin(q4).bwd_x(g, dy2, BT)
        self._lin(q2).wgrad(dy2, ops.src(y1), BT, Yact=y2)
        self._lin(q2).bwd_x(dy2, dy1, BT, Yact=y2)
        self._lin(q0).wgrad(dy1, ops.src(s, esum), BT, Yact=y1)
        Lin(q0.weight.data[:, S:], None, self._bf16()).bwd_x(dy1, desum, BT, Yact=y1)     # only the enc columns need a gradient
        ops.agent_bcast(desum, de2, BT, N, ae)
        x_ha = ops.src(hidden, idx=u_idx.view(R, 1), nhot=1, hot_w=A)
        self._lin(e2).wgrad(de2, ops.src(e1), R)
        self._lin(e2).bwd_x(de2, de1, R)
        self._lin(e0).wgrad(de1, x_ha, R, Yact=e1)
        Lin(e0.weight.data[:, :H], None, self._bf16()).bwd_x(de1, dhidden, R, Yact=e1, beta=1.0 if accumulate else 0.0)

    def forward(self, state, hidden_states, actions):
        dev = require_cuda("QtranQBase")
        self.to(dev)
        B, T, N, A = actions.shape
        s = to_dev(state, dev).reshape(B * T, -1)
        h = to_dev(hidden_states, dev).reshape(B * T * N, -1)
        u_idx = onehot_to_index(to_dev(actions, dev)).reshape(-1)
        return self.hip_forward(s, h, u_idx, B * T, tag="f").clone().view(B * T, 1)


class QtranQAlt(nn.Module):
    """Name kept for import compatibility (reference network/mixer.py:295-351).  The reference's
    qtran_alt path raises at run time (out of scope), so this is not implemented."""

    def __init__(self, args):
        super().__init__()
        raise NotImplementedError("qtran_alt is broken in the reference and is not part of the hot path")


class QtranV(_QtranFusedHead, _Precision, nn.Module):
    """QTRAN state-value network (reference network/mixer.py:392-418)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        H = args.rnn_hidden_dim
        self.hidden_encoding = _mlp(None, [H, H, H])
        self.v = _mlp(None, [args.state_shape + H, args.qtran_hidden_dim, args.qtran_hidden_dim, 1])
        self._s = _Scratch()

    def _qt_layers(self):
        return tuple(_linears(self.hidden_encoding) + _linears(self.v))

    def _qt_actions(self):
        return 0

    def hip_forward(self, s, hidden, BT, ctx=None, tag="e", sp=None):
        if self._qt_ok(hidden):
            return self._qt_forward(s, hidden, None, BT, ctx, tag, sp)
        a = self.args
        N, H, Q = a.n_agents, a.rnn_hidden_dim, a.qtran_hidden_dim
        R = BT * N
        dev = s.device
        e0, e2 = _linears(self.hidden_encoding)
        v0, v2, v4 = _linears(self.v)
        e1 = self._s.get("e1" + tag, (R, H), dev)
        e2b = self._s.get("e2" + tag, (R, H), dev)
        esum = self._s.get("esum" + tag, (BT, H), dev)
        y1 = self._s.get("y1" + tag, (BT, Q), dev)
        y2 = self._s.get("y2" + tag, (BT, Q), dev)
        out = self._s.get("out" + tag, (BT, 1), dev)
        self._lin(e0).fwd(ops.src(hidden), e1, R, act=1)
        self._lin(e2).fwd(ops.src(e1), e2b, R)
        ops.agent_sum(e2b, esum, BT, N, H)
        self._lin(v0).fwd(ops.src(s, esum), y1, BT, act=1)
        self._lin(v2).fwd(ops.src(y1), y2, BT, act=1)
        self._lin(v4).fwd(ops.src(y2), out, BT)
        if ctx is not None:
            ctx.update(s=s, hidden=hidden, e1=e1, esum=esum, y1=y1, y2=y2)
        return out.view(BT)

    def hip_backward(self, ctx, d_out, BT, dhidden, accumulate):
        if ctx.get("fused"):
            return self._qt_backward(ctx, d_out, BT, dhidden, accumulate)
        a = self.args
        N, H, Q, S = a.n_agents, a.rnn_hidden_dim, a.qtran_hidden_dim, a.state_shape
        R = BT * N
        dev = d_out.device
        e0, e2 = _linears(self.hidden_encoding)
        v0, v2, v4 = _linears(self.v)
        s, hidden, e1, esum, y1, y2 = (ctx[k] for k in ("s", "hidden", "e1", "esum", "y1", "y2"))
        g = d_out.view(BT, 1)
        dy2 = self._s.get("dy2", (BT, Q), dev)
        dy1 = self._s.get("dy1", (BT, Q), dev)
        desum = self._s.get("desum", (BT, H), dev)
        de2 = self._s.get("de2", (R, H), dev)
        de1 = self._s.get("de1", (R, H), dev)
        self._lin(v4).wgrad(g, ops.src(y2), BT)
        self._lin(v4).bwd_x(g, dy2, BT)
        self._lin(v2).wgrad(dy2, ops.src(y1), BT, Yact=y2)
        self._lin(v2).bwd_x(dy2, dy1, BT, Yact=y2)
        self._lin(v0).wgrad(dy1, ops.src(s, esum), BT, Yact=y1)
        Lin(v0.weight.data[:, S:], None, self._bf16()).bwd_x(dy1, desum, BT, Yact=y1)
        ops.agent_bcast(desum, de2, BT, N, H)
        self._lin(e2).wgrad(de2, ops.src(e1), R)
        self._lin(e2).bwd_x(de2, de1, R)
        self._lin(e0).wgrad(de1, ops.src(hidden), R, Yact=e1)
        self._lin(e0).bwd_x(de1, dhidden, R, Yact=e1, beta=1.0 if accumulate else 0.0)

    def forward(self, state, hidden):
        dev = require_cuda("QtranV")
        self.to(dev)
        B, T, N, _ = hidden.shape
        s = to_dev(state, dev).reshape(B * T, -1)
        h = to_dev(hidden, dev).reshape(B * T * N, -1)
        return self.hip_forward(s, h, B * T, tag="f").clone().view(B * T, 1)
