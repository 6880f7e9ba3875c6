"""CPU restatement of QLearner.train / QTRANLearner.train (TEST INFRASTRUCTURE).

State is explicit: dicts of tensors for eval/target agent, mixer/target mixer (and V),
plus optimizer state.  Gradients come from torch autograd (third-party arithmetic the
reference uses too); clip-norm and the RMSprop / Adam updates are written out from the
torch formulas.
"""
from __future__ import annotations

import copy
import numpy as np
import torch

from . import nets

MASK_BIG = -9999999.0      # algorithm/q_learner.py:105,112,126 ; qtran_learner.py:106
MASK_QTRAN_EVAL = -999999.0  # algorithm/qtran_learner.py:105


def max_episode_len(terminated, episode_limit):
    """QLearner.get_max_episode_len (algorithm/q_learner.py:49-66), incl. quirk Q2:
    an episode that never terminates is ignored; 0 -> episode_limit."""
    term = np.asarray(terminated)
    B = term.shape[0]
    m = 0
    for b in range(B):
        hits = np.nonzero(term[b, :episode_limit, 0] == 1)[0]
        if hits.size and hits[0] + 1 >= m:
            m = int(hits[0]) + 1
    return m if m > 0 else episode_limit


def to_tensors(batch, T):
    """Slice to [:, :T] and convert (u long, rest float32) as q_learner.py:63-78."""
    out = {}
    for k, v in batch.items():
        v = np.asarray(v)[:, :T]
        out[k] = torch.tensor(v, dtype=torch.long if k == "u" else torch.float32)
    return out


class LearnerState:
    """Everything a learner owns (q_learner.py:11-47 / qtran_learner.py:11-50)."""

    def __init__(self, args, agent, mixer, v=None, extra=None):
        f = lambda d: {k: torch.tensor(np.asarray(x), dtype=torch.float32).clone().requires_grad_(True)
                       for k, x in d.items()}
        self.args = args
        self.agent = f(agent)
        self.mixer = f(mixer)
        self.v = f(v) if v is not None else None
        self.extra = f(extra) if extra is not None else None  # QTRAN's unused q_sum_mixer (Q11)
        self.target_agent = {k: x.detach().clone() for k, x in self.agent.items()}
        self.target_mixer = {k: x.detach().clone() for k, x in self.mixer.items()}
        self.opt = {}   # name -> state tensors
        self.opt_step = 0

    def named_params(self):
        """Order of ``self.params`` in the reference: agent, mixer, (v, q_sum_mixer)."""
        out = [("agent." + k, x) for k, x in self.agent.items()]
        out += [("mixer." + k, x) for k, x in self.mixer.items()]
        if self.v is not None:
            out += [("v." + k, x) for k, x in self.v.items()]
        if self.extra is not None:
            out += [("q_sum_mixer." + k, x) for k, x in self.extra.items()]
        return out

    def sync_targets(self):
        """_update_targets (q_learner.py:181-184)."""
        self.target_agent = {k: x.detach().clone() for k, x in self.agent.items()}
        self.target_mixer = {k: x.detach().clone() for k, x in self.mixer.items()}


def clip_and_step(state: LearnerState, grads: dict):
    """clip_grad_norm_(params, clip) then RMSprop/Adam with torch defaults
    (q_learner.py:42-47,170-173; quirk Q12: RMSprop alpha 0.99, eps 1e-8)."""
    args = state.args
    live = [g for g in grads.values() if g is not None]
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in live)).float() if live else torch.tensor(0.0)
    coef = min(1.0, float(args.grad_norm_clip) / (float(total) + 1e-6))
    state.opt_step += 1
    t = state.opt_step
    with torch.no_grad():
        for name, p in state.named_params():
            g = grads.get(name)
            if g is None:
                continue
            g = g * coef
            if args.optimizer == "RMS":
                sq = state.opt.setdefault(name, torch.zeros_like(p))
                sq.mul_(0.99).addcmul_(g, g, value=0.01)
                p.addcdiv_(g, sq.sqrt().add_(1e-8), value=-args.lr)
            elif args.optimizer == "Adam":
                m, v = state.opt.setdefault(name, (torch.zeros_like(p), torch.zeros_like(p)))
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
                denom = (v.sqrt() / (bc2 ** 0.5)).add_(1e-8)
                p.addcdiv_(m, denom, value=-args.lr / bc1)
            else:
                raise ValueError("optimizer {} not recognised.".format(args.optimizer))
    return float(total), coef


# ---------------------------------------------------------------------------------
# QLearner.train  (algorithm/q_learner.py:68-179)
# ---------------------------------------------------------------------------------
def q_forward(state: LearnerState, batch, want=None, T=None):
    """Forward of the VDN/QMIX/QPLEX loss.  Returns (loss, dict of intermediates).
    ``T`` overrides get_max_episode_len (data-parallel shards must agree on it)."""
    args = state.args
    if T is None:
        T = max_episode_len(batch["terminated"], args.episode_limit)
    bt = to_tensors(batch, T)
    B, N, H = bt["o"].shape[0], args.n_agents, args.rnn_hidden_dim
    s, u, r, s_next = bt["s"], bt["u"], bt["r"], bt["s_next"]
    avail_u, avail_next, term, u_onehot = bt["avail_u"], bt["avail_u_next"], bt["terminated"], bt["u_onehot"]
    mask = 1.0 - bt["padded"]
    la, ru = args.last_action, args.reuse_network

    h0 = torch.zeros(B * N, H)
    q_evals, hs_eval, h_last = nets.agent_unroll(state.agent, bt["o"], nets.shifted_onehot(u_onehot), h0, la, ru)
    q_chosen = torch.gather(q_evals, 3, u).squeeze(3)

    with torch.no_grad():
        q_tgt, hs_tgt, _ = nets.agent_unroll(state.target_agent, bt["o_next"], u_onehot, h0, la, ru)
        q_tgt = q_tgt.clone()
        q_tgt[avail_next == 0.0] = MASK_BIG
        if args.double_q:
            # quirk Q1: continues from the eval net's final hidden state, no re-init (q_learner.py:96-110)
            q_en, _, _ = nets.agent_unroll(state.agent, bt["o_next"], u_onehot, h_last.detach(), la, ru)
            q_en = q_en.clone()
            q_en[avail_next == 0] = MASK_BIG
            cur_max = q_en.argmax(dim=3, keepdim=True)
            q_tgt_chosen = torch.gather(q_tgt, 3, cur_max).squeeze(3)
        else:
            cur_max = None
            q_tgt_chosen = q_tgt.max(dim=3)[0]

    inter = dict(T=T, q_evals=q_evals, hs_eval=hs_eval, q_targets=q_tgt, q_targets_chosen=q_tgt_chosen)
    if args.alg == "qplex":
        v_tot = nets.qplex(state.mixer, q_chosen, s, args, is_v=True)
        qd = q_evals.detach().clone()
        qd[avail_u == 0] = MASK_BIG
        max_q = qd.max(dim=3)[0]
        a_tot = nets.qplex(state.mixer, q_chosen, s, args, actions=u_onehot, max_q_i=max_q, is_v=False)
        q_tot = v_tot + a_tot
        with torch.no_grad():
            if args.double_q:
                onehot = torch.zeros_like(u_onehot).scatter_(3, cur_max, 1)
                vt = nets.qplex(state.target_mixer, q_tgt_chosen, s_next, args, is_v=True)
                q_tgt_max = q_tgt.max(dim=3)[0]
                at = nets.qplex(state.target_mixer, q_tgt_chosen, s_next, args, actions=onehot,
                                max_q_i=q_tgt_max, is_v=False)
                q_tot_tgt = vt + at
            else:
                q_tot_tgt = nets.qplex(state.target_mixer, q_tgt_chosen, s_next, args, is_v=True)
        inter.update(v_tot=v_tot, a_tot=a_tot)
    elif args.alg == "qmix":
        q_tot = nets.qmix(state.mixer, q_chosen, s, args)
        with torch.no_grad():
            q_tot_tgt = nets.qmix(state.target_mixer, q_tgt_chosen, s_next, args)
    elif args.alg == "vdn":
        q_tot = nets.vdn(q_chosen)
        q_tot_tgt = nets.vdn(q_tgt_chosen)
    else:
        raise ValueError("Mixer {} not recognised.".format(args.alg))

    targets = r + args.gamma * q_tot_tgt * (1 - term)
    td = targets.detach() - q_tot
    mtd = mask * td
    loss = (mtd ** 2).sum() / mask.sum()
    inter.update(q_tot=q_tot, q_tot_target=q_tot_tgt, loss=loss, num=(mtd ** 2).sum(), den=mask.sum())
    return loss, inter


def _grads(state, loss):
    named = state.named_params()
    gs = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
    return {n: g for (n, _), g in zip(named, gs)}


def q_train(state: LearnerState, batch, train_step):
    """One QLearner.train call; returns (loss float, grads-before-clip, intermediates)."""
    loss, inter = q_forward(state, batch)
    grads = _grads(state, loss)
    norm, coef = clip_and_step(state, grads)
    if train_step > 0 and train_step % state.args.target_update_cycle == 0:
        state.sync_targets()
    inter.update(grad_norm=norm, clip_coef=coef)
    return float(loss.detach()), grads, inter


# ---------------------------------------------------------------------------------
# QTRANLearner.train  (algorithm/qtran_learner.py:71-163, get_qtran :165-200)
# ---------------------------------------------------------------------------------
def qtran_forward(state: LearnerState, batch, T=None):
    args = state.args
    if T is None:
        T = max_episode_len(batch["terminated"], args.episode_limit)
    bt = to_tensors(batch, T)
    B, N, H = bt["o"].shape[0], args.n_agents, args.rnn_hidden_dim
    s, u, r, s_next = bt["s"], bt["u"], bt["r"], bt["s_next"]
    avail_u, avail_next, term, u_onehot = bt["avail_u"], bt["avail_u_next"], bt["terminated"], bt["u_onehot"]
    mask = 1.0 - bt["padded"].squeeze(-1)
    la, ru = args.last_action, args.reuse_network

    h0 = torch.zeros(B * N, H)
    q_ind, hs_eval, _ = nets.agent_unroll(state.agent, bt["o"], nets.shifted_onehot(u_onehot), h0, la, ru)
    with torch.no_grad():
        q_ind_tgt, hs_tgt, _ = nets.agent_unroll(state.target_agent, bt["o_next"], u_onehot, h0, la, ru)
        q_ind_tgt = q_ind_tgt.clone()
        q_ind_tgt[avail_next == 0.0] = MASK_BIG
        opt_tgt = torch.zeros_like(q_ind_tgt).scatter(-1, q_ind_tgt.argmax(dim=3, keepdim=True), 1)
    q_clone = q_ind.clone()
    q_clone[avail_u == 0.0] = MASK_QTRAN_EVAL
    opt_eval = torch.zeros_like(q_clone).scatter(-1, q_clone.argmax(dim=3, keepdim=True), 1).detach()

    joint_q = nets.qtran_q(state.mixer, s, hs_eval, u_onehot, args).view(B, -1)
    with torch.no_grad():
        joint_q_tgt = nets.qtran_q(state.target_mixer, s_next, hs_tgt, opt_tgt, args).view(B, -1)
    v = nets.qtran_v(state.v, s, hs_eval, args).view(B, -1)

    y = r.squeeze(-1) + args.gamma * joint_q_tgt * (1 - term.squeeze(-1))
    l_td = (((joint_q - y.detach()) * mask) ** 2).sum() / mask.sum()

    q_sum_opt = q_clone.max(dim=-1)[0].sum(dim=-1)
    joint_q_hat = nets.qtran_q(state.mixer, s, hs_eval, opt_eval, args).view(B, -1)
    l_opt = (((q_sum_opt - joint_q_hat.detach() + v) * mask) ** 2).sum() / mask.sum()

    q_sum_nopt = torch.gather(q_ind, -1, u).squeeze(-1).sum(dim=-1)
    nopt = (q_sum_nopt - joint_q.detach() + v).clamp(max=0)
    l_nopt = ((nopt * mask) ** 2).sum() / mask.sum()

    loss = l_td + args.lambda_opt * l_opt + args.lambda_nopt * l_nopt
    inter = dict(T=T, q_evals=q_ind, hs_eval=hs_eval, q_targets=q_ind_tgt, hs_target=hs_tgt,
                 joint_q_evals=joint_q, joint_q_targets=joint_q_tgt, v=v, joint_q_hat_opt=joint_q_hat,
                 l_td=l_td, l_opt=l_opt, l_nopt=l_nopt, loss=loss, den=mask.sum())
    return loss, inter


def qtran_train(state: LearnerState, batch, train_step):
    loss, inter = qtran_forward(state, batch)
    grads = _grads(state, loss)
    norm, coef = clip_and_step(state, grads)
    if train_step > 0 and train_step % state.args.target_update_cycle == 0:
        state.sync_targets()
    inter.update(grad_norm=norm, clip_coef=coef)
    return float(loss.detach()), grads, inter


def train(state, batch, train_step):
    if state.args.alg.startswith("qtran"):
        return qtran_train(state, batch, train_step)
    return q_train(state, batch, train_step)


def clone_batch(batch):
    return {k: copy.deepcopy(v) for k, v in batch.items()}
