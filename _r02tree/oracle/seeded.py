"""Deterministic weights / batches shared by the golden-vector generator and the tests.

TEST INFRASTRUCTURE.  numpy PCG64 streams are stable across platforms, so neither the
weights nor the inputs need to be stored in the fixtures - only a checksum of them.

Shapes follow the BASELINE.json configs (SURVEY.md section 8 "Config shapes").
"""
from __future__ import annotations

import types
import numpy as np

# name -> dims.  T here is the SMAC episode_limit; tests use tiny B,T but real N/O/S/A.
SHAPES = {
    "matrix": dict(n_agents=2, obs_shape=1, state_shape=1, n_actions=3, episode_limit=1),
    "2s3z": dict(n_agents=5, obs_shape=80, state_shape=120, n_actions=11, episode_limit=120),
    "3s5z": dict(n_agents=8, obs_shape=128, state_shape=216, n_actions=14, episode_limit=150),
    "MMM2": dict(n_agents=10, obs_shape=176, state_shape=322, n_actions=18, episode_limit=120),
}


def make_args(shape: str, alg: str, episode_limit: int | None = None, **over):
    """Namespace with every field the path reads (reference common/arguments.py:88-146)."""
    a = types.SimpleNamespace()
    a.__dict__.update(SHAPES[shape])
    if episode_limit is not None:
        a.episode_limit = episode_limit
    a.alg = alg
    a.map = shape
    a.rnn_hidden_dim = 64
    a.qmix_hidden_dim = 32
    a.two_hyper_layers = False
    a.hyper_hidden_dim = 64
    a.qtran_hidden_dim = 64
    a.lr = 5e-4
    a.epsilon = 1.0
    a.min_epsilon = 0.05
    a.anneal_epsilon = (a.epsilon - a.min_epsilon) / 50000
    a.epsilon_anneal_scale = "step"
    a.train_steps = 1
    a.batch_size = 32
    a.buffer_size = 5000
    a.save_cycle = 5000
    a.target_update_cycle = 200
    a.lambda_opt = 1
    a.lambda_nopt = 1
    a.grad_norm_clip = 10
    a.adv_hypernet_embed = 64
    a.num_kernel = 10
    a.adv_hypernet_layers = 3
    a.weighted_head = True
    a.hypernet_embed = 64
    a.is_minus_one = True
    a.mixing_embed_dim = 32
    a.double_q = True
    a.last_action = True
    a.reuse_network = True
    a.gamma = 0.99
    a.optimizer = "RMS"
    a.cuda = False
    a.RTW = False
    a.load_model = False
    a.model_dir = "./model"
    a.result_dir = "./result"
    a.replay_dir = ""
    a.n_episodes = 1
    a.evaluate_epoch = 0
    a.evaluate_cycle = 5000
    a.n_steps = 800000
    a.env = "smac"
    a.seed = 123
    for k, v in over.items():
        setattr(a, k, v)
    return a


# ---------------------------------------------------------------------------------
# parameter shapes, in torch state_dict order of the reference modules
# ---------------------------------------------------------------------------------
def agent_param_shapes(args):
    """RNNQNet (network/q_network.py:9-14); input dim per share_params.py:114-123."""
    H, A = args.rnn_hidden_dim, args.n_actions
    I = args.obs_shape + (A if args.last_action else 0) + (args.n_agents if args.reuse_network else 0)
    return [("fc1.weight", (H, I)), ("fc1.bias", (H,)),
            ("rnn.weight_ih", (3 * H, H)), ("rnn.weight_hh", (3 * H, H)),
            ("rnn.bias_ih", (3 * H,)), ("rnn.bias_hh", (3 * H,)),
            ("fc2.weight", (A, H)), ("fc2.bias", (A,))]


def _lin(prefix, out_f, in_f):
    return [(prefix + ".weight", (out_f, in_f)), (prefix + ".bias", (out_f,))]


def qmix_param_shapes(args):
    """QMixMixer (network/mixer.py:30-55)."""
    S, N, E, HH = args.state_shape, args.n_agents, args.qmix_hidden_dim, args.hyper_hidden_dim
    out = []
    if args.two_hyper_layers:
        out += _lin("hyper_w1.0", HH, S) + _lin("hyper_w1.2", N * E, HH)
        out += _lin("hyper_w2.0", HH, S) + _lin("hyper_w2.2", E, HH)
    else:
        out += _lin("hyper_w1", N * E, S) + _lin("hyper_w2", E, S)
    out += _lin("hyper_b1", E, S)
    out += _lin("hyper_b2.0", E, S) + _lin("hyper_b2.2", 1, E)
    return out


def qplex_param_shapes(args):
    """DMAQer + DMAQ_SI_Weight (network/mixer.py:85-147, 184-209), adv_hypernet_layers=3."""
    S, N, A = args.state_shape, args.n_agents, args.n_actions
    HE, AE, K = args.hypernet_embed, args.adv_hypernet_embed, args.num_kernel
    assert args.adv_hypernet_layers == 3
    out = _lin("hyper_w_final.0", HE, S) + _lin("hyper_w_final.2", N, HE)
    out += _lin("V.0", HE, S) + _lin("V.2", N, HE)
    for k in range(K):
        p = "si_weight.key_extractors.%d" % k
        out += _lin(p + ".0", AE, S) + _lin(p + ".2", AE, AE) + _lin(p + ".4", 1, AE)
    for k in range(K):
        p = "si_weight.agents_extractors.%d" % k
        out += _lin(p + ".0", AE, S) + _lin(p + ".2", AE, AE) + _lin(p + ".4", N, AE)
    for k in range(K):
        p = "si_weight.action_extractors.%d" % k
        out += _lin(p + ".0", AE, S + N * A) + _lin(p + ".2", AE, AE) + _lin(p + ".4", N, AE)
    return out


def qtran_q_param_shapes(args):
    """QtranQBase (network/mixer.py:360-375)."""
    H, A, S, Q = args.rnn_hidden_dim, args.n_actions, args.state_shape, args.qtran_hidden_dim
    ae = H + A
    return (_lin("hidden_action_encoding.0", ae, ae) + _lin("hidden_action_encoding.2", ae, ae)
            + _lin("q.0", Q, S + A + H) + _lin("q.2", Q, Q) + _lin("q.4", 1, Q))


def qtran_v_param_shapes(args):
    """QtranV (network/mixer.py:397-409)."""
    H, S, Q = args.rnn_hidden_dim, args.state_shape, args.qtran_hidden_dim
    return (_lin("hidden_encoding.0", H, H) + _lin("hidden_encoding.2", H, H)
            + _lin("v.0", Q, S + H) + _lin("v.2", Q, Q) + _lin("v.4", 1, Q))


def mixer_param_shapes(args):
    return {"vdn": lambda a: [], "qmix": qmix_param_shapes, "qplex": qplex_param_shapes,
            "qtran_base": qtran_q_param_shapes}[args.alg](args)


def seeded_state(shapes, seed, scale=1.0):
    """torch-default-like init (U(+-1/sqrt(fan_in)); GRU U(+-1/sqrt(H))) from a numpy stream."""
    rng = np.random.default_rng(seed)
    out = {}
    fan = {}
    for name, shp in shapes:
        base = name.rsplit(".", 1)[0]
        if name.endswith("weight") and len(shp) == 2:
            fan[base] = shp[1]
    for name, shp in shapes:
        base = name.rsplit(".", 1)[0]
        if base == "rnn" or name.startswith("rnn."):
            bound = 1.0 / np.sqrt(shp[-1] if len(shp) == 2 else shp[0] // 3)
        else:
            bound = 1.0 / np.sqrt(fan[base])
        out[name] = (rng.uniform(-bound, bound, size=shp) * scale).astype(np.float32)
    return out


# ---------------------------------------------------------------------------------
# synthetic batches: the 11-key episode dict (reference rollout.py:135-146)
# ---------------------------------------------------------------------------------
def make_batch(args, B, seed, lengths=None, p_avail=0.7, dtype=np.float64, full_length=False):
    """Seeded episode batch with ragged lengths, padding and unavailable actions.

    Layout/dtypes as the reference rollout emits: o,s,r,u_onehot,padded,
    terminated float; u int64 (B,T,N,1); avail_* float.  Padded steps are all-zero with
    padded=1, terminated=1 (rollout.py:122-133).  ``lengths[b] = -1`` makes an episode that
    runs the full episode_limit WITHOUT terminating (quirk Q2).
    """
    rng = np.random.default_rng(seed)
    T, N, O, S, A = args.episode_limit, args.n_agents, args.obs_shape, args.state_shape, args.n_actions
    if lengths is None:
        if full_length:
            lengths = [T] * B
        else:
            lengths = list(rng.integers(max(1, T // 2), T + 1, size=B))
    obs = rng.standard_normal((B, T + 1, N, O))
    st = rng.standard_normal((B, T + 1, S))
    avail = (rng.random((B, T + 1, N, A)) < p_avail).astype(np.float64)
    avail[..., 0] = 1.0
    u = np.zeros((B, T, N, 1), dtype=np.int64)
    u_onehot = np.zeros((B, T, N, A))
    r = rng.standard_normal((B, T, 1))
    padded = np.zeros((B, T, 1))
    term = np.zeros((B, T, 1))
    pick = rng.random((B, T, N))
    for b in range(B):
        L = T if lengths[b] < 0 else int(lengths[b])
        for t in range(T):
            if t >= L:
                padded[b, t, 0] = 1.0
                term[b, t, 0] = 1.0
                obs[b, t + 1] = 0.0
                st[b, t + 1] = 0.0
                avail[b, t + 1] = 0.0
                r[b, t, 0] = 0.0
                continue
            for n in range(N):
                idx = np.nonzero(avail[b, t, n])[0]
                a = idx[int(pick[b, t, n] * len(idx))]
                u[b, t, n, 0] = a
                u_onehot[b, t, n, a] = 1.0
            if t == L - 1 and lengths[b] >= 0:
                term[b, t, 0] = 1.0
    # obs[L] / avail[L] stay real: they are o_next / avail_u_next of the last real step
    o, o_next = obs[:, :-1].copy(), obs[:, 1:].copy()
    s, s_next = st[:, :-1].copy(), st[:, 1:].copy()
    av, av_next = avail[:, :-1].copy(), avail[:, 1:].copy()
    for b in range(B):
        L = T if lengths[b] < 0 else int(lengths[b])
        o[b, L:] = 0.0
        s[b, L:] = 0.0
        av[b, L:] = 0.0
        o_next[b, L:] = 0.0
        s_next[b, L:] = 0.0
        av_next[b, L:] = 0.0
    batch = dict(o=o, s=s, u=u, r=r, avail_u=av, o_next=o_next, s_next=s_next,
                 avail_u_next=av_next, u_onehot=u_onehot, padded=padded, terminated=term)
    for k in batch:
        if k != "u":
            batch[k] = batch[k].astype(dtype)
    return batch


def checksum(arrs) -> float:
    """Order-sensitive float64 checksum of a list/dict of arrays (fixture drift guard)."""
    if isinstance(arrs, dict):
        arrs = [arrs[k] for k in sorted(arrs)]
    tot = 0.0
    for i, a in enumerate(arrs):
        a = np.asarray(a, dtype=np.float64).ravel()
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.37 + i)
        tot += float(np.dot(a, w))
    return tot


def sample_indices(n, k=64):
    """Fixed strided subset used to pin large tensors (grads / params) in fixtures."""
    if n <= k:
        return np.arange(n)
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))
