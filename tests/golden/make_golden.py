#!/usr/bin/env python3
"""Generate the golden vectors by running the REAL reference (import from /root/reference).

Run in the build container only (the reference never travels):
    python tests/golden/make_golden.py
Writes small ``.npz`` fixtures next to this file.  Inputs and weights are NOT stored: they
are regenerated from numpy PCG64 seeds by ``oracle/seeded.py`` (a checksum of them is
stored to detect drift).  Large tensors (gradients, post-step parameters) are pinned by
their per-tensor L2 norm, sum and a fixed strided sample (``seeded.sample_indices``).
"""
import os
import sys
import types
import copy

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MARL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.argv = ["x"]
np.float = float      # shims the reference needs on numpy >= 1.24
np.long = int
sys.modules.setdefault("gym", types.SimpleNamespace(Env=object))

from oracle import seeded, rollout as orl  # noqa: E402

from controller.share_params import SharedMAC  # noqa: E402  (reference)
from algorithm.q_learner import QLearner  # noqa: E402
from algorithm.qtran_learner import QTRANLearner  # noqa: E402
from common.replaybuffer import ReplayBuffer  # noqa: E402
from rollout import RolloutWorker  # noqa: E402
from env.single_state_matrix_game import TwoAgentsMatrixGame  # noqa: E402
import network.mixer as ref_mixer  # noqa: E402

th.set_num_threads(1)

# (name, shape, alg, B, T(episode_limit), lengths, overrides)
CASES = [
    ("vdn_matrix", "matrix", "vdn", 32, 1, [1] * 32, {}),
    ("qmix_2s3z", "2s3z", "qmix", 4, 6, [5, 3, -1, 4], {}),
    ("qmix_2s3z_adam", "2s3z", "qmix", 3, 5, [5, 2, 4], {"optimizer": "Adam"}),
    ("qmix_2s3z_hyper2", "2s3z", "qmix", 3, 5, [-1, -1, -1], {"two_hyper_layers": True}),
    ("vdn_2s3z_nodq", "2s3z", "vdn", 3, 5, [3, 5, 4], {"double_q": False}),
    ("qplex_2s3z", "2s3z", "qplex", 4, 6, [6, 3, -1, 4], {}),
    ("qplex_2s3z_nodq", "2s3z", "qplex", 3, 4, [4, 2, 3], {"double_q": False}),
    ("qtran_3s5z", "3s5z", "qtran_base", 4, 6, [6, 2, -1, 5], {}),
    ("qmix_MMM2", "MMM2", "qmix", 3, 5, [5, 3, 4], {}),
    # round 3: the shapes that left the marl_linear composition (heads with 322 / 502 input columns, 320-wide hypernet heads)
    ("qplex_MMM2", "MMM2", "qplex", 3, 4, [4, 2, 3], {}),
    ("qmix_MMM2_hyper2", "MMM2", "qmix", 3, 5, [5, 2, 4], {"two_hyper_layers": True}),
    ("qplex_3s5z", "3s5z", "qplex", 3, 4, [3, -1, 4], {}),
]
TRAIN_STEPS = [0, 1, 200, 201]   # 200 crosses the target-sync boundary (quirk Q6)


def load(module, state):
    module.load_state_dict({k: th.tensor(v) for k, v in state.items()})


def pin(prefix, named, out):
    """norm / sum / strided sample per tensor."""
    for name, t in named:
        if t is None:
            out["%s/%s/none" % (prefix, name)] = np.array(1)
            continue
        a = t.detach().cpu().numpy().astype(np.float64).ravel()
        out["%s/%s/norm" % (prefix, name)] = np.array(np.sqrt((a * a).sum()))
        out["%s/%s/sum" % (prefix, name)] = np.array(a.sum())
        out["%s/%s/samp" % (prefix, name)] = a[seeded.sample_indices(a.size)].astype(np.float32)


def named_params(learner):
    out = [("agent." + k, p) for k, p in learner.eval_net.agent.named_parameters()]
    out += [("mixer." + k, p) for k, p in learner.mixer.named_parameters()]
    if hasattr(learner, "v"):
        out += [("v." + k, p) for k, p in learner.v.named_parameters()]
        out += [("q_sum_mixer." + k, p) for k, p in learner.q_sum_mixer.named_parameters()]
    return out


def build(case):
    name, shape, alg, B, T, lengths, over = case
    args = seeded.make_args(shape, alg, episode_limit=T, **over)
    mac = SharedMAC(args)
    load(mac.agent, seeded.seeded_state(seeded.agent_param_shapes(args), seed=11))
    learner = QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args)
    mshapes = seeded.mixer_param_shapes(args)
    if mshapes:
        st = seeded.seeded_state(mshapes, seed=12)
        load(learner.mixer, st)
        load(learner.target_mixer, st)
    if alg.startswith("qtran"):
        load(learner.v, seeded.seeded_state(seeded.qtran_v_param_shapes(args), seed=13))
        load(learner.q_sum_mixer, seeded.seeded_state(seeded.qmix_param_shapes(args), seed=14))
    return args, mac, learner


def gen_learner_case(case):
    name, shape, alg, B, T, lengths, over = case
    args, mac, learner = build(case)
    out = {"meta/B": np.array(B), "meta/T": np.array(T), "meta/lengths": np.array(lengths)}

    # ---- standalone forward pieces on batch seed 100 (before any update)
    batch = seeded.make_batch(args, B, seed=100, lengths=lengths)
    out["meta/batch_checksum"] = np.array(seeded.checksum(batch))
    tb = {k: th.tensor(v, dtype=th.long if k == "u" else th.float32) for k, v in batch.items()}
    with th.no_grad():
        mac.init_hidden(B)
        q_cur, h_cur = mac.get_current_q_values(tb, T)
        q_nxt_cont, h_nxt_cont = mac.get_next_q_values(tb, T)      # continues from final hidden (Q1)
        mac.init_hidden(B)
        q_nxt, h_nxt = mac.get_next_q_values(tb, T)
        out["fwd/q_cur"], out["fwd/h_cur"] = q_cur.numpy(), h_cur.numpy()
        out["fwd/q_next"], out["fwd/h_next"] = q_nxt.numpy(), h_nxt.numpy()
        out["fwd/q_next_cont"] = q_nxt_cont.numpy()
        qc = th.gather(q_cur, 3, tb["u"]).squeeze(3)
        if alg in ("vdn", "qmix"):
            out["fwd/q_tot"] = learner.mixer(qc, tb["s"]).numpy()
        elif alg == "qplex":
            qd = q_cur.clone(); qd[tb["avail_u"] == 0] = -9999999
            mx = qd.max(dim=3)[0]
            out["fwd/v_tot"] = learner.mixer(qc, tb["s"], is_v=True).numpy()
            out["fwd/a_tot"] = learner.mixer(qc, tb["s"], actions=tb["u_onehot"], max_q_i=mx, is_v=False).numpy()
            out["fwd/lambda"] = learner.mixer.si_weight(tb["s"], tb["u_onehot"]).numpy()
        else:
            out["fwd/joint_q"] = learner.mixer(tb["s"], h_cur, tb["u_onehot"]).numpy()
            out["fwd/v"] = learner.v(tb["s"], h_cur).numpy()

    # ---- train steps, capturing pre-clip grads
    captured = {}
    orig_clip = th.nn.utils.clip_grad_norm_

    def spy(params, max_norm, *a, **k):
        params = list(params)
        captured["grads"] = [None if p.grad is None else p.grad.detach().clone() for p in params]
        captured["norm"] = orig_clip(params, max_norm, *a, **k)
        return captured["norm"]

    th.nn.utils.clip_grad_norm_ = spy
    try:
        losses = []
        for i, ts in enumerate(TRAIN_STEPS):
            b = seeded.make_batch(args, B, seed=100 + i, lengths=lengths)
            losses.append(learner.train(b, ts))
            names = [n for n, _ in named_params(learner)]
            assert len(names) == len(captured["grads"])
            pin("step%d/grad" % i, list(zip(names, captured["grads"])), out)
            out["step%d/grad_norm" % i] = np.array(float(captured["norm"]))
            pin("step%d/param" % i, named_params(learner), out)
            pin("step%d/target_agent" % i, [("agent." + k, p) for k, p in learner.target_net.agent.named_parameters()], out)
        out["losses"] = np.array(losses, dtype=np.float64)
        out["meta/T_used"] = np.array(learner.max_episode_len)
    finally:
        th.nn.utils.clip_grad_norm_ = orig_clip
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "losses", losses)


def gen_rollout():
    out = {}
    # config 1: VDN matrix game, 32 episodes, reference RolloutWorker with its numpy RNG
    args = seeded.make_args("matrix", "vdn")
    mac = SharedMAC(args)
    load(mac.agent, seeded.seeded_state(seeded.agent_param_shapes(args), seed=11))
    env = TwoAgentsMatrixGame([[8, -12, -12], [-12, 0, 0], [-12, 0, 0]])
    for tag, eps in (("eps1", 1.0), ("eps03", 0.3)):
        args.epsilon = eps
        w = RolloutWorker(env, mac, args)
        np.random.seed(7)
        ep, rew, wins, steps = w.generate_episodes(32)
        for k, v in ep.items():
            out["matrix_%s/%s" % (tag, k)] = np.asarray(v, dtype=np.float64)
        out["matrix_%s/rewards" % tag] = np.array(rew, dtype=np.float64)
        out["matrix_%s/steps" % tag] = np.array(steps)
        out["matrix_%s/eps_after" % tag] = np.array(w.epsilon)
    ge = env.get_episodes()
    for k, v in ge.items():
        out["matrix_get_episodes/" + k] = np.asarray(v, dtype=np.float64)

    # SMAC-shaped serial rollout on the synthetic env (2s3z dims, T=8)
    args = seeded.make_args("2s3z", "qmix", episode_limit=8)
    mac = SharedMAC(args)
    load(mac.agent, seeded.seeded_state(seeded.agent_param_shapes(args), seed=11, scale=3.0))
    for tag, eps, evaluate in (("greedy", 0.0, True), ("eps05", 0.5, False)):
        sy = orl.SynthSMAC(5, 80, 120, 11, 8, seed=5)
        env = orl.SerialSynthEnv(sy)
        args.epsilon = eps
        w = RolloutWorker(env, mac, args)
        np.random.seed(9)
        ep, rew, wins, steps = w.generate_episodes(6, evaluate=evaluate)
        for k in ("u", "r", "padded", "terminated", "avail_u", "avail_u_next"):
            out["smac_%s/%s" % (tag, k)] = np.asarray(ep[k], dtype=np.float64)
        out["smac_%s/o_checksum" % tag] = np.array(seeded.checksum([ep["o"], ep["o_next"], ep["s"], ep["s_next"]]))
        out["smac_%s/rewards" % tag] = np.array(rew, dtype=np.float64)
        out["smac_%s/wins" % tag] = np.array(wins)
        out["smac_%s/steps" % tag] = np.array(steps)
        out["smac_%s/eps_after" % tag] = np.array(w.epsilon)
    np.savez_compressed(os.path.join(HERE, "rollout.npz"), **out)
    print("rollout fixtures written")


def gen_replay():
    out = {}
    args = seeded.make_args("2s3z", "qmix", episode_limit=3, buffer_size=7)
    buf = ReplayBuffer(args)
    sizes = [1, 3, 2, 3, 1, 7, 2]
    idx_log, state_log = [], []
    rng = np.random.default_rng(3)
    for i, n in enumerate(sizes):
        ep = seeded.make_batch(args, n, seed=300 + i)
        before = buf.current_idx
        buf.store_episode(ep)
        state_log.append([before, buf.current_idx, buf.current_size])
    out["state_log"] = np.array(state_log)
    np.random.seed(21)
    s = buf.sample(5)
    out["sample_r"] = s["r"]
    out["sample_u"] = s["u"]
    out["final_r"] = buf.buffers["r"].copy()
    np.savez_compressed(os.path.join(HERE, "replay.npz"), **out)
    print("replay fixtures written")


def gen_matrix_table():
    """get_q_and_q_tot_table (q_learner.py:211-262, qtran_learner.py:237-272) on seeded nets."""
    out = {}
    for alg in ("vdn", "qmix", "qplex", "qtran_base"):
        case = ("x", "matrix", alg, 9, 1, [1] * 9, {})
        args, mac, learner = build(case)
        qt, qi, qj = learner.get_q_and_q_tot_table()
        out[alg + "/q_tot"], out[alg + "/q_i"], out[alg + "/q_j"] = qt, qi, qj
    np.savez_compressed(os.path.join(HERE, "matrix_table.npz"), **out)
    print("matrix table fixtures written")


if __name__ == "__main__":
    only = os.environ.get("MARL_GOLDEN_ONLY", "")      # comma-separated case names: (re)generate just these learner cases
    for c in CASES:
        if not only or c[0] in only.split(","):
            gen_learner_case(c)
    if not only:
        gen_rollout()
        gen_replay()
        gen_matrix_table()
