"""Functional CPU restatement of the agent network and the mixers.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters are plain dicts
name -> torch tensor using the reference's state_dict key names; every function is
differentiable through torch autograd so the learner oracle can take gradients.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def lin(p, prefix, x):
    """y = x W^T + b for the nn.Linear stored under ``prefix``."""
    return F.linear(x, p[prefix + ".weight"], p[prefix + ".bias"])


class _LinBf16(torch.autograd.Function):
    """y = r(x) r(W)^T + b with both GEMM operands rounded to bf16 and fp32 accumulation - the build's opt-in "bf16 mixer"
    mode (BASELINE config 5), restated here so that the oracle can check it: products of bf16 values are exact in fp32, so only
    the accumulation order differs from the matrix cores.  Backward as the build does it: dW = r(g)^T r(x) - the weight-gradient
    GEMM takes bf16 operands too (round 4) -, db = colsum(g) in fp32, no gradient into x (states); the gradient that flows on
    into the mixing arithmetic (and from there into the agents) is formed from the forward's values in fp32."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x)
        return F.linear(x.bfloat16().float(), W.bfloat16().float(), b)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return None, g.bfloat16().float().t() @ x.bfloat16().float(), g.sum(0)


def lin_bf16(p, prefix, x):
    return _LinBf16.apply(x, p[prefix + ".weight"], p[prefix + ".bias"])


# ---------------------------------------------------------------------------------
# agent: RNNQNet (reference network/q_network.py:16-21; GRUCell gate order r,z,n)
# ---------------------------------------------------------------------------------
def agent_step(p, inp, h):
    """inp (rows,I), h (rows,H) -> q (rows,A), h' (rows,H).  SURVEY App. A.1."""
    H = h.shape[-1]
    x = torch.relu(lin(p, "fc1", inp))
    gi = F.linear(x, p["rnn.weight_ih"], p["rnn.bias_ih"])
    gh = F.linear(h, p["rnn.weight_hh"], p["rnn.bias_hh"])
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    h2 = (1.0 - z) * n + z * h
    q = lin(p, "fc2", h2)
    return q, h2


def build_inputs(obs_t, prev_onehot_t, n_agents, last_action=True, reuse_network=True):
    """[obs || one-hot of the action fed back || agent id], flattened to (B*N, I).

    Follows controller/share_params.py:84-112: ``prev_onehot_t`` is zeros at t=0 for the
    current-Q pass and u_onehot[:, t] for the next-Q pass."""
    B = obs_t.shape[0]
    parts = [obs_t]
    if last_action:
        parts.append(prev_onehot_t)
    if reuse_network:
        parts.append(torch.eye(n_agents, dtype=obs_t.dtype).unsqueeze(0).expand(B, -1, -1))
    return torch.cat([x.reshape(B * n_agents, -1) for x in parts], dim=1)


def agent_unroll(p, obs, fed_onehot, h0, last_action=True, reuse_network=True):
    """T-step unroll (share_params.py:125-146 / 148-168).

    obs (B,T,N,O); fed_onehot (B,T,N,A) = the one-hot action concatenated at step t
    (already shifted by the caller); h0 (B*N,H).
    Returns q (B,T,N,A), hs (B,T,N,H) = hidden AFTER each step, h_last (B*N,H)."""
    B, T, N, _ = obs.shape
    h = h0
    qs, hs = [], []
    for t in range(T):
        inp = build_inputs(obs[:, t], fed_onehot[:, t], N, last_action, reuse_network)
        q, h = agent_step(p, inp, h)
        qs.append(q.view(B, N, -1))
        hs.append(h.view(B, N, -1))
    return torch.stack(qs, 1), torch.stack(hs, 1), h


def shifted_onehot(u_onehot):
    """One-hot fed to the CURRENT-Q pass: zeros at t=0, u_onehot[t-1] after (share_params.py:96-100)."""
    return torch.cat([torch.zeros_like(u_onehot[:, :1]), u_onehot[:, :-1]], dim=1)


# ---------------------------------------------------------------------------------
# mixers
# ---------------------------------------------------------------------------------
def vdn(q_chosen):
    """VDNMixer.forward (network/mixer.py:15-16): (B,T,N) -> (B,T,1)."""
    return q_chosen.sum(dim=2, keepdim=True)


def qmix(p, q_chosen, states, args):
    """QMixMixer.forward (network/mixer.py:57-80).  SURVEY App. A.2."""
    B = q_chosen.shape[0]
    N, E, S = args.n_agents, args.qmix_hidden_dim, args.state_shape
    qv = q_chosen.reshape(-1, N)
    s = states.reshape(-1, S)
    # mixer_dtype == "bf16" (build extension, not in the reference): the four state-conditioned hypernet GEMMs of the
    # single-layer hypernets take bf16 operands; everything else stays fp32
    slin = lin_bf16 if (getattr(args, "mixer_dtype", "fp32") == "bf16" and not args.two_hyper_layers) else lin
    if args.two_hyper_layers:
        w1 = lin(p, "hyper_w1.2", torch.relu(lin(p, "hyper_w1.0", s)))
        w2 = lin(p, "hyper_w2.2", torch.relu(lin(p, "hyper_w2.0", s)))
    else:
        w1 = slin(p, "hyper_w1", s)
        w2 = slin(p, "hyper_w2", s)
    w1 = w1.abs().view(-1, N, E)                       # agent-major: flat index n*E+e
    b1 = slin(p, "hyper_b1", s)
    hid = F.elu((qv.unsqueeze(2) * w1).sum(1) + b1)    # (rows,E)
    w2 = w2.abs()
    b2 = lin(p, "hyper_b2.2", torch.relu(slin(p, "hyper_b2.0", s)))  # (rows,1)
    q_tot = (hid * w2).sum(1, keepdim=True) + b2
    return q_tot.view(B, -1, 1)


def _mlp3(p, prefix, x):
    x = torch.relu(lin(p, prefix + ".0", x))
    x = torch.relu(lin(p, prefix + ".2", x))
    return lin(p, prefix + ".4", x)


def qplex_lambda(p, states, actions_onehot, args):
    """DMAQ_SI_Weight.forward (network/mixer.py:149-171): (rows,S),(rows,N*A) -> (rows,N)."""
    S, N, A = args.state_shape, args.n_agents, args.n_actions
    s = states.reshape(-1, S)
    a = actions_onehot.reshape(-1, N * A)
    sa = torch.cat([s, a], dim=1)
    lam = 0.0
    for k in range(args.num_kernel):
        key = _mlp3(p, "si_weight.key_extractors.%d" % k, s)        # (rows,1)
        ag = _mlp3(p, "si_weight.agents_extractors.%d" % k, s)      # (rows,N)
        ac = _mlp3(p, "si_weight.action_extractors.%d" % k, sa)     # (rows,N)
        lam = lam + (key.abs() + 1e-10) * torch.sigmoid(ag) * torch.sigmoid(ac)
    return lam


def qplex_transform(p, states, args):
    """|W(s)|+1e-10 and V(s) of the transformation net (network/mixer.py:262-267)."""
    s = states.reshape(-1, args.state_shape)
    w = lin(p, "hyper_w_final.2", torch.relu(lin(p, "hyper_w_final.0", s))).abs() + 1e-10
    v = lin(p, "V.2", torch.relu(lin(p, "V.0", s)))
    return w, v


def qplex(p, agent_qs, states, args, actions=None, max_q_i=None, is_v=False):
    """DMAQer.forward (network/mixer.py:251-288) with weighted_head / is_minus_one flags."""
    B = agent_qs.shape[0]
    N = args.n_agents
    w, v = qplex_transform(p, states, args)
    q = agent_qs.reshape(-1, N)
    if args.weighted_head:
        q = w * q + v
    if is_v:
        y = q.sum(-1)
    else:
        m = max_q_i.reshape(-1, N)
        if args.weighted_head:
            m = w * m + v
        adv = (q - m).detach()
        lam = qplex_lambda(p, states, actions, args)
        y = (adv * (lam - 1.0)).sum(1) if args.is_minus_one else (adv * lam).sum(1)
    return y.view(B, -1, 1)


def qtran_q(p, states, hidden, actions_onehot, args):
    """QtranQBase.forward (network/mixer.py:378-388): -> (B*T,1).  SURVEY App. A.4."""
    B, T, N, _ = actions_onehot.shape
    ha = torch.cat([hidden, actions_onehot], dim=-1).reshape(B * T * N, -1)
    enc = lin(p, "hidden_action_encoding.2", torch.relu(lin(p, "hidden_action_encoding.0", ha)))
    enc = enc.view(B * T, N, -1).sum(1)
    x = torch.cat([states.reshape(B * T, -1), enc], dim=-1)
    return _mlp3(p, "q", x)


def qtran_v(p, states, hidden, args):
    """QtranV.forward (network/mixer.py:411-418): -> (B*T,1)."""
    B, T, N, H = hidden.shape
    enc = lin(p, "hidden_encoding.2", torch.relu(lin(p, "hidden_encoding.0", hidden.reshape(-1, H))))
    enc = enc.view(B * T, N, -1).sum(1)
    x = torch.cat([states.reshape(B * T, -1), enc], dim=-1)
    return _mlp3(p, "v", x)
