"""Thin Python wrappers over the C ABI: torch tensors in, raw device pointers out.

torch is used only for device memory and the stream handle (plumbing); every computation
below is a hand-written HIP kernel from marl_amd/csrc.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (MarlSrc, MarlGroup, MarlAgentWeights, MarlAgentGrads, MarlQmixWeights, MarlMlp3Weights,
                   MarlQtranWeights, check)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, "expected a CUDA float32 tensor"
    return t


def _i32(t):
    assert t.dtype == torch.int32 and t.is_cuda, "expected a CUDA int32 tensor"
    return t


class Workspace:
    """Grow-only scratch buffers keyed by name (no allocation in steady state).  ``gen`` counts (re)allocations:
    a captured hipGraph holds raw pointers into these buffers, so its owner records ``gen`` at capture time, keeps
    ``snapshot()`` alive (the old storage cannot be freed and handed to somebody else under the graph) and drops
    the graph when ``gen`` has moved (algorithm/common.py:GraphedUpdate)."""

    def __init__(self):
        self.bufs = {}
        self.gen = 0

    def get(self, name, nbytes, device):
        n = (int(nbytes) + 3) // 4
        b = self.bufs.get(name)
        if b is None or b.numel() < n or b.device != device:
            b = torch.empty(max(n, 1), dtype=torch.float32, device=device)
            self.bufs[name] = b
            self.gen += 1
        return b

    def snapshot(self):
        return list(self.bufs.values())


WS = Workspace()


class Rows:
    """A 2-D row source read through a row remap (rpe, bs, off) - e.g. the (T+1)-slot state storage - and
    optionally an int32 episode map (replay samples read in place from the ring)."""

    def __init__(self, t, remap, emap=None):
        self.t, self.remap, self.emap = t, remap, emap
        self.device = t.device


def src(x0=None, x1=None, idx=None, nhot=0, hot_w=0, nid=0, gate=None, remap0=None, remapi=None, k0=None, emap0=None):
    """Build a marl_src_t.  x0/x1: 2-D (rows, k) float32 views with unit inner stride (x0 may be Rows)."""
    s = MarlSrc()
    keep = []
    if isinstance(x0, Rows):
        x0, remap0, emap0 = x0.t, x0.remap, x0.emap
    if x0 is not None:
        _f32(x0); assert x0.dim() == 2 and x0.stride(1) == 1
        s.p0, s.ld0, s.k0 = x0.data_ptr(), x0.stride(0), (x0.shape[1] if k0 is None else k0)
        keep.append(x0)
    if x1 is not None:
        _f32(x1); assert x1.dim() == 2 and x1.stride(1) == 1
        s.p1, s.ld1, s.k1 = x1.data_ptr(), x1.stride(0), x1.shape[1]
        keep.append(x1)
    if idx is not None:
        _i32(idx)
        s.idx, s.nhot, s.hot_w = idx.data_ptr(), nhot, hot_w
        keep.append(idx)
    s.nid = nid
    if gate is not None:
        _f32(gate); assert gate.dim() == 2 and gate.stride(1) == 1
        s.m0, s.ldm0 = gate.data_ptr(), gate.stride(0)
        keep.append(gate)
    if remap0 is not None:
        s.rpe0, s.bs0, s.off0 = remap0
    if emap0 is not None:
        assert remap0 is not None and emap0.dtype == torch.int32 and emap0.is_contiguous()
        s.emap0 = emap0.data_ptr()
        keep.append(emap0)
    if remapi is not None:
        s.rpei, s.bsi, s.offi = remapi
    s._keep = keep
    return s


def src_width(s):
    return s.k0 + s.k1 + s.nhot * s.hot_w + s.nid


def group(groups, x0=0, x1=0, w=0, b=0, y=0, m0=0):
    return MarlGroup(groups, x0, x1, w, b, y, m0)


def linear(x, W, bias, Y, M, N, K, act=0, beta=0.0, w_kmajor=False, ldw=None, grp=None, bf16=False):
    """Y[M,N] = act(X W^T + b) (+beta*Y).  W: (N,K) view, or (K,N) when w_kmajor.  bf16: round both operands to
    bf16 and use the bf16 matrix cores with fp32 accumulation (per call; the mixers pass their args.mixer_dtype)."""
    lib = _lib.load()
    if bf16:
        act |= 0x100
    if ldw is None:
        ldw = W.stride(0) if W.dim() == 2 else (N if w_kmajor else K)
    ldy = Y.stride(0) if Y.dim() == 2 else N
    assert src_width(x) == K, (src_width(x), K)
    check(lib.marl_linear(C.byref(x), _p(_f32(W)), ldw, 1 if w_kmajor else 0, _p(bias), _p(_f32(Y)), ldy,
                          M, N, K, act, float(beta), C.byref(grp) if grp is not None else None, _stream()),
          "marl_linear")


def linear_wgrad(dY, x, dW, db, M, N, K, Yact=None, grp=None, lddw=None, bf16=False):
    """dW[N,K] += (dY * (Yact>0))^T X ; db[N] += column sums."""
    lib = _lib.load()
    flags = 1 if bf16 else 0
    g = grp.groups if grp is not None else 1
    nbytes = lib.marl_linear_wgrad_workspace(M, N, K, g)
    ws = WS.get("wgrad", nbytes, dY.device)
    if lddw is None:
        lddw = dW.stride(0) if dW.dim() == 2 else K
    assert src_width(x) == K, (src_width(x), K)
    check(lib.marl_linear_wgrad(_p(_f32(dY)), dY.stride(0) if dY.dim() == 2 else N,
                                _p(Yact), (Yact.stride(0) if Yact.dim() == 2 else N) if Yact is not None else 0,
                                C.byref(x), _p(_f32(dW)), lddw, _p(db), M, N, K, flags,
                                C.byref(grp) if grp is not None else None, _p(ws), ws.numel() * 4, _stream()),
          "marl_linear_wgrad")


def agent_weights(params):
    """params: dict name -> tensor with RNNQNet keys."""
    w = MarlAgentWeights()
    w.fc1_w, w.fc1_b = params["fc1.weight"].data_ptr(), params["fc1.bias"].data_ptr()
    w.w_ih, w.w_hh = params["rnn.weight_ih"].data_ptr(), params["rnn.weight_hh"].data_ptr()
    w.b_ih, w.b_hh = params["rnn.bias_ih"].data_ptr(), params["rnn.bias_hh"].data_ptr()
    w.fc2_w, w.fc2_b = params["fc2.weight"].data_ptr(), params["fc2.bias"].data_ptr()
    w.H = params["rnn.weight_hh"].shape[1]
    for k in ("fc1.weight", "rnn.weight_ih", "rnn.weight_hh", "fc2.weight"):
        assert params[k].is_contiguous() and params[k].dtype == torch.float32 and params[k].is_cuda
    w._keep = params
    return w


def agent_unroll_fwd(w, obs, obs_bs, obs_t0, ufed, u_bs, u_t0, h0, q, hs, h_last, saved,
                     B, T, N, O, A, last_action=True, reuse_network=True, ep_len=None, ep_map=None, cu_budget=0, gi_out=None,
                     gi_in=None):
    """cu_budget: CUs (= workgroups) a T > 1 launch spreads its rows over (0 / 256 = the whole chip); 128 lets two
    independent unrolls run side by side on two streams (PairedUnroll).  A per-call argument, no process state.
    gi_out / gi_in: (T, B*N, 3, 64) buffer of input-side gate sums written by one unroll and read by a later unroll of the
    same weights whose step t input is the earlier one's step t+1 input (include/marl_hip.h)."""
    lib = _lib.load()
    check(lib.marl_agent_unroll_fwd(C.byref(w), _p(_f32(obs)), obs_bs, obs_t0, _p(ufed), u_bs, u_t0, _p(ep_len),
                                    _p(_i32(ep_map)) if ep_map is not None else None, _p(h0),
                                    _p(_f32(q)), _p(hs), _p(h_last), _p(saved), B, T, N, O, A,
                                    1 if last_action else 0, 1 if reuse_network else 0, int(cu_budget),
                                    _p(_f32(gi_out)) if gi_out is not None else None,
                                    _p(_f32(gi_in)) if gi_in is not None else None, _stream()),
          "marl_agent_unroll_fwd")


def agent_unroll_x6_supported(B, T, N, O, A, last_action=True, reuse_network=True):
    return bool(_lib.load().marl_agent_unroll_x6_supported(B, T, N, O, A, 1 if last_action else 0, 1 if reuse_network else 0))


def agent_unroll_x6_plain_r6(B, T, N, O, A, last_action=True, reuse_network=True, cu_budget=0):
    """a non-saving split unroll of this batch runs on csrc/agent_x6p.hip (the round-6 decomposition): no gate sums are kept for it"""
    return bool(_lib.load().marl_agent_unroll_x6_plain_r6(B, T, N, O, A, 1 if last_action else 0, 1 if reuse_network else 0, int(cu_budget)))


def agent_unroll_fwd_x6(w, obs, obs_bs, obs_t0, ufed, u_bs, u_t0, h0, q, hs, h_last, saved, B, T, N, O, A, last_action=True,
                        reuse_network=True, ep_len=None, ep_map=None, cu_budget=0, gi_out=None, gi_in=None):
    """the unroll on the bf16x6 split kernels (csrc/agent_x6.hip; opt-in args.gemm_mode = "bf16x6"): same arguments as
    agent_unroll_fwd, same `saved` layout (the fp32 BPTT kernel reads it); gi_out / gi_in hold plain sums here, so a storing and a
    reading launch must both be this one"""
    if saved is None and gi_in is None and hs is None and agent_unroll_x6_plain_r6(B, T, N, O, A, last_action, reuse_network, cu_budget):
        # csrc/agent_x6p.hip addresses the observations and the fed actions with 32-bit byte offsets against a uniform base
        assert obs.numel() < 2 ** 30 and (ufed is None or ufed.numel() < 2 ** 30), "observation storage beyond 4 GB: set MARL_UNROLL_R6=0"
    check(_lib.load().marl_agent_unroll_fwd_x6(C.byref(w), _p(_f32(obs)), obs_bs, obs_t0, _p(ufed), u_bs, u_t0, _p(ep_len),
                                               _p(_i32(ep_map)) if ep_map is not None else None, _p(h0), _p(_f32(q)), _p(hs),
                                               _p(h_last), _p(saved), B, T, N, O, A, 1 if last_action else 0,
                                               1 if reuse_network else 0, int(cu_budget),
                                               _p(_f32(gi_out)) if gi_out is not None else None,
                                               _p(_f32(gi_in)) if gi_in is not None else None, _stream()),
          "marl_agent_unroll_fwd_x6")


def saved_shape(T, B, N, planes=6, H=64):
    """Shape of the activation buffer an unroll saves for BPTT (planes = 6) or of its input-side gate sums (planes = 3):
    the kernels use a tile layout [T][16-row tile][plane][column tile][lane][4] (csrc/agent.hip: sv_off), so the row count is
    rounded up to whole tiles.  The buffer is opaque to the host; `saved_plane` decodes one plane."""
    return (T + (1 if planes == 6 else 0), (B * N + 15) // 16 * 16, planes, H)     # (+ the hidden state after the last step)


def saved_plane(saved, plane, rows):
    """(T, rows, 64) view-copy of one plane of a buffer in the tile layout (tests / debugging)."""
    T, R16, P, H = saved.shape
    t = saved.reshape(T, R16 // 16, P, 4, 4, 16, 4)[:, :, plane]          # (T, tile, c, q, m, i)
    return t.permute(0, 1, 3, 5, 2, 4).reshape(T, R16, H)[:, :rows]       # row = 16 tile + 4 q + i ; column = 16 c + m  (T+1 slabs for 6 planes)


def agent_unroll_reuse_supported(B, T, N, O, A, cu_budget=0):
    return bool(_lib.load().marl_agent_unroll_reuse_supported(B, T, N, O, A, int(cu_budget)))


def agent_unroll_bwd_x6_supported(B, T, N, A, sparse_dq=True):
    return bool(_lib.load().marl_agent_unroll_bwd_x6_supported(B, T, N, A, 1 if sparse_dq else 0))


def agent_unroll_bwd(w, dq, dhs, saved, hs, dxp, dh0, grads, B, T, N, A, dq_idx=None, dq_val=None, dq_idx2=None,
                     dq_val2=None, dq_gdiv=1, x6=False):
    """grads: dict name -> gradient tensor for rnn.weight_ih/hh, rnn.bias_ih/hh, fc2.weight/bias (accumulated)."""
    lib = _lib.load()
    g = MarlAgentGrads()
    g.w_ih, g.w_hh = grads["rnn.weight_ih"].data_ptr(), grads["rnn.weight_hh"].data_ptr()
    g.b_ih, g.b_hh = grads["rnn.bias_ih"].data_ptr(), grads["rnn.bias_hh"].data_ptr()
    g.fc2_w, g.fc2_b = grads["fc2.weight"].data_ptr(), grads["fc2.bias"].data_ptr()
    for v in grads.values():
        assert v.is_contiguous() and v.dtype == torch.float32 and v.is_cuda
    if x6:      # the split kernels (csrc/agent_bwd_x6.hip; opt-in args.gemm_mode = "bf16x6"): sparse dq only
        assert dq is None and dq_idx is not None and dq_val is not None and dq_idx.is_contiguous() and dq_val.is_contiguous()
        _i32(dq_idx); _f32(dq_val)
        if dq_idx2 is not None:
            assert dq_val2 is not None and dq_idx2.is_contiguous() and dq_val2.is_contiguous()
            _i32(dq_idx2); _f32(dq_val2)
        ws = WS.get("agent_bwd_x6", lib.marl_agent_bwd_x6_workspace(B, N, A), saved.device)
        check(lib.marl_agent_unroll_bwd_x6(C.byref(w), _p(dq_idx), _p(dq_val), _p(dq_idx2), _p(dq_val2), int(dq_gdiv), _p(dhs),
                                           _p(_f32(saved)), _p(_f32(dxp)), _p(dh0), C.byref(g), _p(ws), ws.numel() * 4, B, T, N, A,
                                           _stream()), "marl_agent_unroll_bwd_x6")
        return
    ws = WS.get("agent_bwd", lib.marl_agent_bwd_workspace(B, N, A), saved.device)
    if dq_idx is not None:
        assert dq is None and dq_val is not None and dq_idx.is_contiguous() and dq_val.is_contiguous()
        _i32(dq_idx); _f32(dq_val)
    if dq_idx2 is not None:
        assert dq_idx is not None and dq_val2 is not None and dq_idx2.is_contiguous() and dq_val2.is_contiguous()
        _i32(dq_idx2); _f32(dq_val2)
    check(lib.marl_agent_unroll_bwd(C.byref(w), _p(_f32(dq)) if dq is not None else None, _p(dq_idx), _p(dq_val),
                                    _p(dq_idx2), _p(dq_val2), int(dq_gdiv), _p(dhs),
                                    _p(_f32(saved)), _p(hs), _p(_f32(dxp)),
                                    _p(dh0), C.byref(g), _p(ws), ws.numel() * 4, B, T, N, A, _stream()),
          "marl_agent_unroll_bwd")


def q_gather(q, idx, out, rows, A, avail=None, mask_val=0.0):
    check(_lib.load().marl_q_gather(_p(_f32(q)), _p(_i32(idx)), _p(avail), float(mask_val), _p(_f32(out)), rows, A,
                                    _stream()), "marl_q_gather")


def q_masked_max(q, avail, mask_val, out_max, out_arg, rows, A):
    check(_lib.load().marl_q_masked_max(_p(_f32(q)), _p(avail), float(mask_val), _p(out_max), _p(out_arg), rows, A,
                                        _stream()), "marl_q_masked_max")


def q_double_select(q_sel, q_val, avail, mask_val, out_val, out_arg, rows, A):
    check(_lib.load().marl_q_double_select(_p(_f32(q_sel)), _p(_f32(q_val)), _p(avail), float(mask_val), _p(_f32(out_val)),
                                           _p(out_arg), rows, A, _stream()), "marl_q_double_select")


def q_scatter(dq, idx1, g1, idx2, g2, rows, A, gdiv=1):
    check(_lib.load().marl_q_scatter(_p(_f32(dq)), _p(idx1), _p(g1), _p(idx2), _p(g2), rows, A, gdiv, _stream()),
          "marl_q_scatter")


def vec_add(a, b, out, n):
    check(_lib.load().marl_vec_add(_p(_f32(a)), _p(_f32(b)), _p(_f32(out)), n, _stream()), "marl_vec_add")


def _ld(t, D):
    """row stride of a (rows, D) view (padded intermediates), D for flat / differently shaped tensors"""
    return t.stride(0) if (t.dim() == 2 and t.shape[1] == D and t.stride(1) == 1) else D


def agent_sum(inp, out, rows, N, D):
    check(_lib.load().marl_agent_sum(_p(_f32(inp)), _ld(inp, D), _p(_f32(out)), _ld(out, D), rows, N, D, _stream()),
          "marl_agent_sum")


def agent_bcast(inp, out, rows, N, D, accumulate=False):
    check(_lib.load().marl_agent_bcast(_p(_f32(inp)), _ld(inp, D), _p(_f32(out)), _ld(out, D), rows, N, D,
                                       1 if accumulate else 0, _stream()), "marl_agent_bcast")


def qmix_mix_fwd(hy, b2, q, q_tot, rows, N, E, w22=None, b22=None):
    """b2 None: formed in the kernel from hy's relu'd fourth block with hyper_b2.2's weight w22 (E) and bias b22 (1)"""
    check(_lib.load().marl_qmix_mix_fwd(_p(_f32(hy)), hy.stride(0), _p(b2), _p(w22), _p(b22), _p(_f32(q)), _p(_f32(q_tot)), rows,
                                        N, E, _stream()), "marl_qmix_mix_fwd")


def qmix_mix_bwd(hy, q, dq_tot, dhy, db2, dq, rows, N, E, w22=None):
    """w22 given: the fourth block of dhy (d hb through hyper_b2.2 and the relu) is written here too"""
    assert dhy.stride(0) == hy.stride(0)
    check(_lib.load().marl_qmix_mix_bwd(_p(_f32(hy)), hy.stride(0), _p(_f32(q)), _p(_f32(dq_tot)), _p(w22), _p(_f32(dhy)),
                                        _p(_f32(db2)), _p(_f32(dq)), rows, N, E, _stream()), "marl_qmix_mix_bwd")


def qmix_tail_supported(S, E, s):
    return E == 32 and qtran_state_parts_supported(S, s)


def qmix_tail_fwd(s, rows, S, b1_w, b1_b, h_w, h_b, hy, c_b1, c_h):
    """hy[:, c_b1:c_b1+32] = hyper_b1(s), hy[:, c_h:c_h+32] = relu(hyper_b2.0(s)) in one pass over s (Rows or dense)"""
    assert b1_w.stride(1) == 1 and h_w.stride(1) == 1 and b1_w.stride(0) == h_w.stride(0) and hy.stride(1) == 1
    x = src(s)
    check(_lib.load().marl_qmix_tail_fwd(C.byref(x), rows, S, _p(_f32(b1_w)), b1_w.stride(0), _p(_f32(b1_b)), _p(_f32(h_w)),
                                         h_w.stride(0), _p(_f32(h_b)), _p(_f32(hy)), hy.stride(0), c_b1, c_h, _stream()),
          "marl_qmix_tail_fwd")


def qplex_mix_fwd(w_raw, v, q, max_q, key, ag, ac, v_tot, a_tot, lam_out, rows, N, K, weighted, minus_one):
    check(_lib.load().marl_qplex_mix_fwd(_p(w_raw), _p(v), _p(q), _p(max_q), _p(key), _p(ag), _p(ac), _p(v_tot),
                                         _p(a_tot), _p(lam_out), rows, N, K, int(weighted), int(minus_one), _stream()),
          "marl_qplex_mix_fwd")


def qplex_mix_bwd(w_raw, q, max_q, key, ag, ac, g, dq, dw_raw, dv, dkey, dag, dac, rows, N, K, weighted, minus_one):
    check(_lib.load().marl_qplex_mix_bwd(_p(w_raw), _p(q), _p(max_q), _p(key), _p(ag), _p(ac), _p(g), _p(dq),
                                         _p(dw_raw), _p(dv), _p(dkey), _p(dag), _p(dac), rows, N, K, int(weighted),
                                         int(minus_one), _stream()), "marl_qplex_mix_bwd")


_FT_OUT = {}


def replay_gather(idx, src, out):
    """src / out: objects with u, r, term, padded, length, won (+ src.avail, out.o_map, out.u_act, out.avail_next and, when the
    learner masks current-step actions, out.avail_cur)."""
    assert idx.dtype == torch.int64 and idx.is_cuda and idx.is_contiguous()
    B, T, N, A = int(idx.numel()), src.T, src.N, src.A
    for t in (src.u, src.r, src.term, src.padded, src.length, src.won, src.avail, out.u, out.u_act, out.r, out.term, out.padded,
              out.length, out.won, out.avail_next, out.o_map):
        assert t.is_cuda and t.is_contiguous()
    assert src.u.dtype == torch.int32 and src.length.dtype == torch.int32 and src.won.dtype == torch.int32
    check(_lib.load().marl_replay_gather(_p(idx), B, T, N, A, _p(src.u), _p(_f32(src.r)), _p(_f32(src.term)), _p(_f32(src.padded)),
                                         _p(src.length), _p(src.won), _p(_f32(src.avail)), _p(_i32(out.o_map)), _p(_i32(out.u)),
                                         _p(_i32(out.u_act)), _p(_f32(out.r)), _p(_f32(out.term)), _p(_f32(out.padded)),
                                         _p(_i32(out.length)), _p(_i32(out.won)), _p(_f32(out.avail_next)),
                                         _p(getattr(out, "avail_cur", None)), _stream()),
          "marl_replay_gather")


def first_terminated_len(term, T):
    """max over episodes of (first terminated step + 1) within the first T steps as a 1-element int32 device tensor
    (0 = no episode terminates).  term: (E, >=T[, 1]) CUDA float32 with unit inner stride."""
    t2 = term.reshape(term.shape[0], -1)
    assert t2.dtype == torch.float32 and t2.is_cuda and t2.stride(1) == 1
    ring = _FT_OUT.get(t2.device)
    if ring is None:         # a ring of output words: an asynchronous read-back of one call may still be pending at the next
        ring = _FT_OUT[t2.device] = [torch.zeros(8, dtype=torch.int32, device=t2.device), 0]
    out = ring[0][ring[1] % 8:ring[1] % 8 + 1]
    ring[1] += 1
    check(_lib.load().marl_first_terminated_len(_p(t2), t2.stride(0), t2.shape[0], min(T, t2.shape[1]), _p(out), _stream()),
          "marl_first_terminated_len")
    return out


def td_loss(q_tot, q_tgt, r, term, padded, gamma, dq_tot, out2, rows):
    lib = _lib.load()
    ws = WS.get("loss", lib.marl_loss_workspace(rows), q_tot.device)
    check(lib.marl_td_loss(_p(_f32(q_tot)), _p(_f32(q_tgt)), _p(_f32(r)), _p(_f32(term)), _p(_f32(padded)),
                           float(gamma), _p(_f32(dq_tot)), _p(_f32(out2)), _p(ws), rows, _stream()), "marl_td_loss")


def qtran_loss(jq, jq_tgt, v, jq_hat, qs_opt, qs_nopt, r, term, padded, gamma, lam_opt, lam_nopt,
               d_jq, d_v, d_qso, d_qsn, out4, rows):
    lib = _lib.load()
    ws = WS.get("loss", lib.marl_loss_workspace(rows), jq.device)
    check(lib.marl_qtran_loss(_p(jq), _p(jq_tgt), _p(v), _p(jq_hat), _p(qs_opt), _p(qs_nopt), _p(r), _p(term),
                              _p(padded), float(gamma), float(lam_opt), float(lam_nopt), _p(d_jq), _p(d_v), _p(d_qso),
                              _p(d_qsn), _p(out4), _p(ws), rows, _stream()), "marl_qtran_loss")


def grad_sumsq(g, n, out1):
    lib = _lib.load()
    ws = WS.get("sumsq", lib.marl_sumsq_workspace(n), g.device)
    check(lib.marl_grad_sumsq(_p(_f32(g)), n, _p(_f32(out1)), _p(ws), _stream()), "marl_grad_sumsq")


def rmsprop_step(p, g, sq, n, lr, alpha, eps, clip, sumsq, den):
    check(_lib.load().marl_rmsprop_step(_p(_f32(p)), _p(_f32(g)), _p(_f32(sq)), n, float(lr), float(alpha), float(eps),
                                        float(clip), _p(sumsq), _p(den), _stream()), "marl_rmsprop_step")


def adam_step(p, g, m, v, n, lr, b1, b2, eps, bc1, bc2s, clip, sumsq, den):
    check(_lib.load().marl_adam_step(_p(_f32(p)), _p(_f32(g)), _p(_f32(m)), _p(_f32(v)), n, float(lr), float(b1),
                                     float(b2), float(eps), float(bc1), float(bc2s), float(clip), _p(sumsq), _p(den),
                                     _stream()), "marl_adam_step")


def select_actions(q, avail, avail_es, alive, eps, rseed, env0, tg, tg0, act_out, act_es, E, N, A):
    check(_lib.load().marl_select_actions(_p(_f32(q)), _p(_f32(avail)), avail_es, _p(alive), float(eps),
                                          int(rseed) & 0xFFFFFFFF, env0, _p(tg), tg0, _p(_i32(act_out)), act_es,
                                          E, N, A, _stream()), "marl_select_actions")


def synth_lengths(seed, env0, episode, length, won, E, T):
    check(_lib.load().marl_synth_lengths(int(seed) & 0xFFFFFFFF, env0, episode, _p(_i32(length)), _p(won), E, T,
                                         _stream()), "marl_synth_lengths")


def synth_observe(seed, env0, episode, t, length, obs, state, avail, E, T, N, O, S, A):
    check(_lib.load().marl_synth_observe(int(seed) & 0xFFFFFFFF, env0, episode, t, _p(_i32(length)), _p(_f32(obs)),
                                         _p(_f32(state)), state.stride(-2), _p(_f32(avail)), E, T, N, O, S, A, _stream()),
          "marl_synth_observe")


def synth_step(seed, env0, episode, t, length, act, u, r, term, padded, alive_next, E, T, N, A):
    check(_lib.load().marl_synth_step(int(seed) & 0xFFFFFFFF, env0, episode, t, _p(_i32(length)), _p(_i32(act)),
                                      _p(_i32(u)), _p(_f32(r)), _p(_f32(term)), _p(_f32(padded)), _p(alive_next),
                                      E, T, N, A, _stream()), "marl_synth_step")


def synth_fused_step(seed, rseed, env0, episode, t, eps, length, q, obs, state, avail, u, r, term, padded, E, T, N, O, S, A):
    check(_lib.load().marl_synth_fused_step(int(seed) & 0xFFFFFFFF, int(rseed) & 0xFFFFFFFF, env0, episode, t, float(eps),
                                            _p(_i32(length)), _p(_f32(q)), _p(_f32(obs)), _p(_f32(state)), state.stride(-2),
                                            _p(_f32(avail)),
                                            _p(_i32(u)), _p(_f32(r)), _p(_f32(term)), _p(_f32(padded)), E, T, N, O, S, A,
                                            _stream()), "marl_synth_fused_step")


def synth_rollout_supported(N, O, A):
    return bool(_lib.load().marl_synth_rollout_supported(N, O, A))


def synth_rollout_x6_supported(N, O, A):
    return bool(_lib.load().marl_synth_rollout_x6_supported(N, O, A))


def synth_rollout_x6_plan(E, N, O, A, last_action=True, reuse_network=True):
    """(decomposition 1 / 2, workgroups, row tiles per workgroup, environments per workgroup, fc1 chunks) of a split whole rollout"""
    plan = (C.c_int * 5)()
    check(_lib.load().marl_synth_rollout_x6_plan(E, N, O, A, 1 if last_action else 0, 1 if reuse_network else 0, plan), "marl_synth_rollout_x6_plan")
    return tuple(plan)


def synth_rollout(w, seed, rseed, env0, episode, fixed_len, eps, rec, h_out, E, T, N, O, S, A, last_action, reuse_network,
                  stats=None, eps_sched=None, x6=False):
    """eps: device (T,) epsilon per lock-step, or None with eps_sched = (eps0, anneal, eps_min): the per-step anneal of
    rollout.py:100-101 is then evaluated inside the kernel (fp64, like the host loop) and no schedule crosses PCIe."""
    e0, ea, em = (0.0, 0.0, 0.0) if eps_sched is None else eps_sched
    assert (eps is None) != (eps_sched is None)
    fn = _lib.load().marl_synth_rollout_x6 if x6 else _lib.load().marl_synth_rollout       # x6: the agent step as bf16x6 split products
    check(fn(C.byref(w), int(seed) & 0xFFFFFFFF, int(rseed) & 0xFFFFFFFF, env0, episode,
                                         1 if fixed_len else 0, _p(_f32(eps)) if eps is not None else None, _p(_f32(rec.obs)), _p(_f32(rec.state)),
                                         rec.state.stride(-2), _p(_f32(rec.avail)), _p(_i32(rec.u)), _p(_f32(rec.r)), _p(_f32(rec.term)),
                                         _p(_f32(rec.padded)), _p(_i32(rec.length)), _p(_i32(rec.won)), _p(h_out),
                                         _p(_f32(stats)) if stats is not None else None, float(e0), float(ea), float(em), E, T, N, O, S, A, 1 if last_action else 0, 1 if reuse_network else 0, _stream()),
          "marl_synth_rollout")


def qmix_fused_supported(N, S, E):
    return bool(_lib.load().marl_qmix_fused_supported(N, S, E))


def qmix_weights(t):
    """t: dict with tensors w1,w1_b,b1,b1_b,w2,w2_b,h,h_b,b2_w,b2_b (weights or their gradients)."""
    w = MarlQmixWeights()
    for k in ("w1", "w1_b", "b1", "b1_b", "w2", "w2_b", "h", "h_b", "b2_w", "b2_b"):
        v = t[k]
        assert v.is_contiguous() and v.dtype == torch.float32 and v.is_cuda
        setattr(w, k, v.data_ptr())
    w._keep = t
    return w


def qmix_fused_fwd(w, s, q, q_tot, rows, N, S, E, x6=False):
    """x6: the bf16x6 split variant of the kernel (args.gemm_mode = "bf16x6"; same arguments)"""
    lib = _lib.load()
    fn = lib.marl_qmix_fused_fwd_x6 if x6 else lib.marl_qmix_fused_fwd
    check(fn(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(q_tot)), rows, N, S, E, _stream()), "marl_qmix_fused_fwd")


def qmix_fused_bwd(w, s, q, dq_tot, dq, grads, rows, N, S, E, x6=False):
    lib = _lib.load()
    ws = WS.get("qmix_fused", lib.marl_qmix_fused_workspace(rows, N, S), q.device)
    fn = lib.marl_qmix_fused_bwd_x6 if x6 else lib.marl_qmix_fused_bwd
    check(fn(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(dq_tot)), _p(_f32(dq)), C.byref(grads),
             _p(ws), ws.numel() * 4, rows, N, S, E, _stream()), "marl_qmix_fused_bwd")


def qmix_fused_loss_bwd(w, s, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq, grads, loss2, rows, N, S, E, x6=False):
    """fused QMIX backward with the TD loss folded in (include/marl_hip.h); loss2: 2-element device view accumulated into"""
    lib = _lib.load()
    ws = WS.get("qmix_fused", lib.marl_qmix_fused_workspace(rows, N, S), q.device)
    check((lib.marl_qmix_fused_loss_bwd_x6 if x6 else lib.marl_qmix_fused_loss_bwd)(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(q_tot_tgt)), _p(_f32(r)), _p(_f32(term)),
                                       _p(_f32(padded)), float(gamma), _p(q_tot), _p(_f32(dq)), C.byref(grads), _p(_f32(loss2)),
                                       _p(ws), ws.numel() * 4, rows, N, S, E, _stream()), "marl_qmix_fused_loss_bwd")


def _wide_flags(bf16, wgrad_bf16):
    """flags of the wide-state QMIX entry points: bit 0 = bf16 operands of the hypernet GEMM, bit 1 = bf16 operands of the
    weight-gradient GEMM too (its own bit: a C-ABI caller that sets only bit 0 keeps fp32 weight gradients)"""
    return (1 if bf16 else 0) | (2 if (bf16 and (wgrad_bf16 is None or wgrad_bf16)) else 0)


def qmix_wide_loss_bwd(w, s, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq, grads, loss2, rows, N, S, E, bf16=False, wgrad_bf16=None):
    """wide-state fused QMIX backward with the TD loss folded in (include/marl_hip.h); wgrad_bf16: None = follow bf16"""
    lib = _lib.load()
    ws = WS.get("qmix_wide", lib.marl_qmix_wide_workspace(rows, N, S, 1), q.device)
    check(lib.marl_qmix_wide_loss_bwd(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(q_tot_tgt)), _p(_f32(r)), _p(_f32(term)),
                                      _p(_f32(padded)), float(gamma), _p(q_tot), _p(_f32(dq)), C.byref(grads), _p(_f32(loss2)),
                                      _p(ws), ws.numel() * 4, rows, N, S, E, _wide_flags(bf16, wgrad_bf16), _stream()), "marl_qmix_wide_loss_bwd")


def qmix_wide_fwd_kernel(rows, N, S, bf16=False):
    """name prefix (rocprofv3 kernel trace) of the forward kernel qmix_wide_fwd launches for this shape"""
    return _lib.load().marl_qmix_wide_fwd_kernel(rows, N, S, 1 if bf16 else 0).decode()


def qmix_wide_supported(N, S, E):
    return bool(_lib.load().marl_qmix_wide_supported(N, S, E))


def qmix_wide_fwd(w, s, q, q_tot, rows, N, S, E, bf16=False):
    lib = _lib.load()
    ws = WS.get("qmix_wide", lib.marl_qmix_wide_workspace(rows, N, S, 0), q.device)
    check(lib.marl_qmix_wide_fwd(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(q_tot)), _p(ws), ws.numel() * 4, rows, N, S, E,
                                 1 if bf16 else 0, _stream()), "marl_qmix_wide_fwd")


def qmix_wide_bwd(w, s, q, dq_tot, dq, grads, rows, N, S, E, bf16=False, wgrad_bf16=None):
    lib = _lib.load()
    ws = WS.get("qmix_wide", lib.marl_qmix_wide_workspace(rows, N, S, 1), q.device)
    check(lib.marl_qmix_wide_bwd(C.byref(w), C.byref(s), _p(_f32(q)), _p(_f32(dq_tot)), _p(_f32(dq)), C.byref(grads), _p(ws),
                                 ws.numel() * 4, rows, N, S, E, _wide_flags(bf16, wgrad_bf16), _stream()), "marl_qmix_wide_bwd")


def _uniform_stride(ts):
    """element stride between consecutive heads' tensors (one flat parameter buffer), None if not uniform."""
    if len(ts) == 1:
        return 0
    d = [ts[i + 1].data_ptr() - ts[i].data_ptr() for i in range(len(ts) - 1)]
    if any(x != d[0] for x in d) or d[0] <= 0 or d[0] % 16 != 0:
        return None
    return d[0] // 4


def mlp3_weights(heads, grad=False):
    """heads: per head the nn.Linear layers of a Linear-ReLU-Linear-ReLU-Linear stack (or the two of a
    Linear-ReLU-Linear stack: w2 = NULL).  Returns the marl_mlp3_weights_t of head 0 + per-tensor head strides, or
    None when the heads are not laid out at constant strides / not 16-byte aligned (the caller then composes
    marl_linear)."""
    pick = (lambda p: p.grad) if grad else (lambda p: p.data)
    w = MarlMlp3Weights()
    keep = []
    names = (("w1", "b1"), ("w2", "b2"), ("w3", "b3")) if len(heads[0]) == 3 else (("w1", "b1"), ("w3", "b3"))
    for li, (wn, bn) in enumerate(names):
        for name, attr in ((wn, "weight"), (bn, "bias")):
            ts = [pick(getattr(h[li], attr)) for h in heads]
            if any(t is None or not t.is_contiguous() or t.dtype != torch.float32 or not t.is_cuda for t in ts):
                return None
            st = _uniform_stride(ts)
            if st is None or ts[0].data_ptr() % 16 != 0:
                return None
            setattr(w, name, ts[0].data_ptr())
            setattr(w, "gs_" + name, st)
            keep.append(ts)
    w._keep = keep
    return w


def mlp3_wide_head(l0, l2, grad=False):
    """A two-layer head Linear(K1, 64) - ReLU - Linear(64, N3) with more outputs than one launch group holds (160), as
    ``groups`` column blocks that share layer 1: returns (marl_mlp3_weights_t, groups, outputs per group) or None."""
    pick = (lambda p: p.grad) if grad else (lambda p: p.data)
    ts = [pick(l0.weight), pick(l0.bias), pick(l2.weight), pick(l2.bias)]
    if any(t is None or not t.is_contiguous() or t.dtype != torch.float32 or not t.is_cuda or t.data_ptr() % 16 for t in ts):
        return None
    N3 = l2.out_features
    G = (N3 + 159) // 160
    if N3 % (4 * G) or l0.out_features != 64 or l2.in_features != 64:
        return None
    w = MarlMlp3Weights()
    w.w1, w.b1, w.w3, w.b3 = (t.data_ptr() for t in ts)
    w.gs_w1 = w.gs_b1 = 0
    w.gs_w3, w.gs_b3 = (N3 // G) * 64, N3 // G
    w._keep = ts
    return w, G, N3 // G


def mlp3_supported(x, K1, H1, H2, N3, groups):
    return bool(_lib.load().marl_mlp3_supported(C.byref(x), K1, H1, H2, N3, groups))


def mlp3_needs_kept(x, K1, N3=1):
    """True when the fused backward of this shape exists only for kept activations (K1 > 192 or more than 16 outputs)."""
    return bool(_lib.load().marl_mlp3_needs_kept(C.byref(x), K1, N3))


def _head_layout(Y, M, N3, groups):
    """(ld, group stride) of the head outputs: (M, groups*N3) with head g in columns [g*N3, (g+1)*N3), or
    (groups, M, N3) with one contiguous (M, N3) block per head."""
    if Y.dim() == 3:
        assert Y.shape == (groups, M, N3) and Y.is_contiguous()
        return N3, M * N3
    assert Y.dim() == 2 and Y.stride(1) == 1 and Y.shape[1] == groups * N3
    return Y.stride(0), N3


def mlp3_x6_supported(x, K1, H1, H2, N3, groups):
    """the bf16x6 split pair (csrc/mlp3_x6.hip) exists for this head shape"""
    return bool(_lib.load().marl_mlp3_x6_supported(C.byref(x), K1, H1, H2, N3, groups))


def mlp3_save_floats(M, three, groups):
    """floats of the kept-activation buffer of one fused head family (layout private to the kernels; the fp32 and the
    bf16x6 pair share it)."""
    return int(_lib.load().marl_mlp3_save_floats(M, 1 if three else 0, groups))


def mlp3_fwd(w, x, Y, M, K1, N3, groups, hsave=None, x6=False):
    """hsave (float32, >= mlp3_save_floats): keep the hidden activations for mlp3_bwd instead of recomputing them.
    x6: the bf16x6 split kernels (fp32-accurate products on the bf16 matrix cores; opt-in args.gemm_mode = "bf16x6")."""
    ld, gs = _head_layout(Y, M, N3, groups)
    assert src_width(x) == K1
    hp, hn = (_p(_f32(hsave)), hsave.numel()) if hsave is not None else (None, 0)
    fn = _lib.load().marl_mlp3_x6_fwd_save if x6 else _lib.load().marl_mlp3_fwd_save
    check(fn(C.byref(w), C.byref(x), _p(_f32(Y)), ld, gs, hp, hn, M, K1, N3, groups, _stream()),
          "marl_mlp3_x6_fwd" if x6 else "marl_mlp3_fwd")


def mlp3_bwd(w, x, dY, grads, M, K1, N3, groups, hsave=None, x6=False):
    lib = _lib.load()
    ld, gs = _head_layout(dY, M, N3, groups)
    assert src_width(x) == K1
    ws = WS.get("mlp3", lib.marl_mlp3_bwd_workspace(M, K1, N3, groups), dY.device)
    hp, hn = (_p(_f32(hsave)), hsave.numel()) if hsave is not None else (None, 0)
    fn = lib.marl_mlp3_x6_bwd_saved if x6 else lib.marl_mlp3_bwd_saved
    check(fn(C.byref(w), C.byref(x), _p(_f32(dY)), ld, gs, C.byref(grads), _p(ws), ws.numel() * 4,
             hp, hn, M, K1, N3, groups, _stream()), "marl_mlp3_x6_bwd" if x6 else "marl_mlp3_bwd")


# ---- fused QTRAN-base heads (csrc/qtran_fused.hip)
def qtran_supported(N, A, AE):
    return bool(_lib.load().marl_qtran_supported(N, A, AE))


def qtran_weights(enc0, enc2, q0, q2, q4, S):
    """nn.Linear layers of hidden(_action)_encoding.{0,2} and q.{0,2,4} (or hidden_encoding / v) -> marl_qtran_weights_t"""
    w = MarlQtranWeights()
    for name, lin in (("enc0", enc0), ("enc2", enc2), ("q2", q2), ("q4", q4)):
        for suf, t in (("_w", lin.weight.data), ("_b", lin.bias.data)):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
            setattr(w, name + suf, t.data_ptr())
    assert q0.weight.data.is_contiguous()
    w.q0_w, w.q0_ld, w.q0_s = q0.weight.data.data_ptr(), q0.weight.shape[1], S
    w._keep = (enc0, enc2, q0, q2, q4)
    return w


def qtran_head_fwd(w, hidden, u, sp, out, s1, e2, y1, y2, BT, N, A, AE):
    check(_lib.load().marl_qtran_head_fwd(C.byref(w), _p(_f32(hidden)), _p(u), _p(_f32(sp)), _p(_f32(out)), _p(s1), _p(e2),
                                          _p(y1), _p(y2), BT, N, A, AE, _stream()), "marl_qtran_head_fwd")


def _state_rows_ok(s):
    t = s.t if isinstance(s, Rows) else s
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.dtype == torch.float32


def qtran_state_parts_supported(S, s=None):
    return bool(_lib.load().marl_qtran_state_parts_supported(S)) and (s is None or _state_rows_ok(s))


def qtran_state_parts(s, BT, S, sets):
    """sets: one or two (weight (64, >= S) row-major, bias (64), out (BT, 64)) triples evaluated on the same rows of s"""
    (W0, b0, o0), (W1, b1, o1) = sets[0], (sets[1] if len(sets) > 1 else (None, None, None))
    for W in (W0, W1):
        assert W is None or (W.stride(1) == 1 and W.dtype == torch.float32)
    x = src(s)
    check(_lib.load().marl_qtran_state_parts(C.byref(x), BT, S, len(sets), _p(W0), W0.stride(0), _p(_f32(b0)), _p(_f32(o0)),
                                             _p(W1), W1.stride(0) if W1 is not None else 0, _p(b1), _p(o1), _stream()),
          "marl_qtran_state_parts")


def qtran_wgrad_rows_supported(S, AE, s=None):
    return bool(_lib.load().marl_qtran_wgrad_rows_supported(S, AE)) and (s is None or _state_rows_ok(s))


def qtran_wgrad_rows(s, s1, e2, y1, y2, d_out, dy1, dy2, de2, d_q0_w, d_q0_b, d_q2_w, d_q2_b, d_q4_w, d_q4_b, d_enc2_w,
                     BT, S, AE):
    lib = _lib.load()
    ws = WS.get("qtran_wg", lib.marl_qtran_wgrad_rows_workspace(S, AE), s.device)
    assert d_q0_w.stride(1) == 1 and d_q2_w.is_contiguous() and d_enc2_w.is_contiguous()
    x = src(s)
    check(lib.marl_qtran_wgrad_rows(C.byref(x), _p(_f32(s1)), _p(_f32(e2)), _p(_f32(y1)), _p(_f32(y2)), _p(_f32(d_out)),
                                    _p(_f32(dy1)), _p(_f32(dy2)), _p(_f32(de2)), _p(d_q0_w), d_q0_w.stride(0), _p(_f32(d_q0_b)),
                                    _p(_f32(d_q2_w)), _p(_f32(d_q2_b)), _p(_f32(d_q4_w)), _p(_f32(d_q4_b)), _p(_f32(d_enc2_w)),
                                    _p(ws), ws.numel() * 4, BT, S, AE, _stream()), "marl_qtran_wgrad_rows")


def qtran_head_fwd2(w, hidden, u, u2, sp, out, out2, s1, e2, y1, y2, BT, N, A, AE):
    check(_lib.load().marl_qtran_head_fwd2(C.byref(w), _p(_f32(hidden)), _p(_i32(u)), _p(_i32(u2)), _p(_f32(sp)), _p(_f32(out)),
                                           _p(_f32(out2)), _p(s1), _p(e2), _p(y1), _p(y2), BT, N, A, AE, _stream()),
          "marl_qtran_head_fwd2")


def qtran_head_bwd(w, hidden, u, d_out, y1, y2, dy1, dy2, de2, dhidden, accumulate, d_enc0_w, d_enc0_b, d_enc2_b,
                   BT, N, A, AE):
    lib = _lib.load()
    ws = WS.get("qtran", lib.marl_qtran_bwd_workspace(BT, AE), hidden.device)
    check(lib.marl_qtran_head_bwd(C.byref(w), _p(_f32(hidden)), _p(u), _p(_f32(d_out)), _p(_f32(y1)), _p(_f32(y2)),
                                  _p(_f32(dy1)), _p(_f32(dy2)), _p(_f32(de2)), _p(_f32(dhidden)), 1 if accumulate else 0,
                                  _p(_f32(d_enc0_w)), _p(_f32(d_enc0_b)), _p(_f32(d_enc2_b)), _p(ws), ws.numel() * 4,
                                  BT, N, A, AE, _stream()), "marl_qtran_head_bwd")
