"""CPU restatement of the rollout side (TEST INFRASTRUCTURE).

* ``serial_rollout``   - RolloutWorker.generate_episodes as the reference runs it
                         (rollout.py:30-173): one env, one agent at a time, numpy global RNG.
* ``batched_rollout``  - the vectorised semantics the HIP path implements: all envs in
                         lock-step, epsilon-greedy draws from the counter hash below.
* ``SynthSMAC``        - the synthetic SMAC-shaped environment (stands in for StarCraft II,
                         which the reference does not vendor): every quantity is a pure
                         function of (seed, env, episode, t, index) through a 32-bit integer
                         hash, so numpy and the HIP kernels agree bit for bit.
* ``MatrixGame``       - env/single_state_matrix_game.py:5-120.
"""
from __future__ import annotations

import numpy as np
import torch

from . import nets

U32 = np.uint32
P_AVAIL = np.float32(0.7)
ST_OBS, ST_STATE, ST_AVAIL, ST_REWARD, ST_LEN, ST_WON, ST_EXPLORE, ST_PICK = range(8)


def mix32(x):
    """lowbias32 integer finaliser on uint32 arrays."""
    x = np.asarray(x, dtype=U32)
    with np.errstate(over="ignore"):
        x = x ^ (x >> U32(16))
        x = x * U32(0x7FEB352D)
        x = x ^ (x >> U32(15))
        x = x * U32(0x846CA68B)
        x = x ^ (x >> U32(16))
    return x


def key(seed, stream, env, t, idx):
    """hash of (seed, stream, env, t, idx) - all broadcastable to uint32 arrays."""
    with np.errstate(over="ignore"):
        h = mix32(U32(seed) + U32(stream) * U32(0x9E3779B1))
        h = mix32(h + np.asarray(env, dtype=U32) * U32(0x85EBCA77) + U32(1))
        h = mix32(h + np.asarray(t, dtype=U32) * U32(0xC2B2AE3D) + U32(2))
        h = mix32(h + np.asarray(idx, dtype=U32) * U32(0x27D4EB2F) + U32(3))
    return h


def u01(h):
    """uint32 -> float32 in [0,1): top 24 bits, exact."""
    return (np.asarray(h, dtype=U32) >> U32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


class SynthSMAC:
    """Synthetic SMAC-shaped env, vectorised over ``env`` ids (numpy restatement)."""

    def __init__(self, n_agents, obs_shape, state_shape, n_actions, episode_limit, seed=1):
        self.N, self.O, self.S, self.A, self.T = n_agents, obs_shape, state_shape, n_actions, episode_limit
        self.seed = seed
        self.lmin = max(1, episode_limit // 2)

    def tg(self, ep, t):
        return np.asarray(ep, dtype=np.int64) * (self.T + 1) + t

    def obs(self, env, ep, t):
        env = np.asarray(env).reshape(-1, 1, 1)
        idx = (np.arange(self.N)[:, None] * self.O + np.arange(self.O)[None, :])[None]
        tg = np.asarray(self.tg(ep, t)).reshape(-1, 1, 1)
        return np.float32(2.0) * u01(key(self.seed, ST_OBS, env, tg, idx)) - np.float32(1.0)

    def state(self, env, ep, t):
        env = np.asarray(env).reshape(-1, 1)
        tg = np.asarray(self.tg(ep, t)).reshape(-1, 1)
        return np.float32(2.0) * u01(key(self.seed, ST_STATE, env, tg, np.arange(self.S)[None])) - np.float32(1.0)

    def avail(self, env, ep, t):
        env = np.asarray(env).reshape(-1, 1, 1)
        idx = (np.arange(self.N)[:, None] * self.A + np.arange(self.A)[None, :])[None]
        tg = np.asarray(self.tg(ep, t)).reshape(-1, 1, 1)
        a = (u01(key(self.seed, ST_AVAIL, env, tg, idx)) < P_AVAIL)
        a[..., 0] = True
        return a.astype(np.float32)

    def reward(self, env, ep, t, actions):
        """actions (E,N) int -> (E,) float32; fixed-order fp32 sum over agents then * (1/N)."""
        env = np.asarray(env).reshape(-1)
        tg = np.asarray(self.tg(ep, t)).reshape(-1)
        acc = np.zeros(env.shape[0], dtype=np.float32)
        for n in range(self.N):
            idx = n * self.A + np.asarray(actions)[:, n]
            acc = acc + (u01(key(self.seed, ST_REWARD, env, tg, idx)) - np.float32(0.5))
        return acc * np.float32(1.0 / self.N)

    def length(self, env, ep):
        h = key(self.seed, ST_LEN, np.asarray(env), np.asarray(ep), 0)
        return (self.lmin + (h % U32(self.T - self.lmin + 1))).astype(np.int64)

    def won(self, env, ep):
        return (key(self.seed, ST_WON, np.asarray(env), np.asarray(ep), 0) & U32(1)).astype(bool)

    def get_env_info(self):
        return dict(n_actions=self.A, n_agents=self.N, state_shape=self.S, obs_shape=self.O,
                    episode_limit=self.T)


class SerialSynthEnv:
    """SMAC-style serial API (rollout.py:37-166 / main.py:16-29) over SynthSMAC.
    The k-th ``reset()`` plays env id k, episode 0 - so n serial episodes equal the
    n parallel env slots of a batched rollout.  With ``env_id`` the object IS env slot ``env_id``
    and its k-th ``reset()`` plays that slot's episode k (n such objects = one batched env)."""

    def __init__(self, synth: SynthSMAC, env_id=None):
        self.sy = synth
        self.fixed = env_id is not None
        self.c = env_id if self.fixed else -1
        self.e = -1 if self.fixed else 0
        self.t = 0

    def reset(self):
        if self.fixed:
            self.e += 1
        else:
            self.c += 1
        self.t = 0
        self.L = int(self.sy.length([self.c], [self.e])[0])

    def get_obs(self):
        return list(self.sy.obs([self.c], [self.e], self.t)[0].astype(np.float64))

    def get_state(self):
        return self.sy.state([self.c], [self.e], self.t)[0].astype(np.float64)

    def get_avail_actions(self):
        return list(self.sy.avail([self.c], [self.e], self.t)[0].astype(np.int64))

    def get_avail_agent_actions(self, i):
        return self.get_avail_actions()[i]

    def step(self, actions):
        a = np.asarray([int(x) for x in actions])[None]
        r = float(self.sy.reward([self.c], [self.e], self.t, a)[0])
        self.t += 1
        done = self.t >= self.L
        info = {"battle_won": bool(self.sy.won([self.c], [self.e])[0])} if done else {}
        return r, done, info

    def get_env_info(self):
        return self.sy.get_env_info()

    def close(self):
        pass

    def save_replay(self):
        pass


class MatrixGame:
    """TwoAgentsMatrixGame (env/single_state_matrix_game.py:5-120)."""

    def __init__(self, payoff):
        self.payoff = np.array(payoff, dtype=np.float64)
        self.n_actions, self.n_agents, self.state_shape, self.obs_shape, self.episode_limit = 3, 2, 1, 1, 1

    def reset(self):
        pass

    def get_obs(self):           # quirk Q9: zeros in rollout ...
        return [np.array([0.0]), np.array([0.0])]

    def get_state(self):
        return np.array([0.0])

    def get_avail_actions(self):
        return [np.array([1, 1, 1]), np.array([1, 1, 1])]

    def get_avail_agent_actions(self, i):
        return np.array([1, 1, 1])

    def step(self, actions):
        return self.payoff[int(actions[0]), int(actions[1])], True, {}

    def close(self):
        pass

    def save_replay(self):
        pass

    def get_env_info(self):
        return dict(n_actions=3, n_agents=2, state_shape=1, obs_shape=1, episode_limit=1)

    def get_episodes(self):      # ... ones here (env/...:81-120), cartesian order (a0,a1)
        n = self.payoff.size
        u = np.zeros((n, 1, 2, 1), dtype=np.int64)
        uo = np.zeros((n, 1, 2, 3))
        for i in range(n):
            a0, a1 = divmod(i, 3)
            u[i, 0, :, 0] = (a0, a1)
            uo[i, 0, 0, a0] = 1
            uo[i, 0, 1, a1] = 1
        one = lambda *s: np.ones(s)
        return dict(o=one(n, 1, 2, 1), s=one(n, 1, 1), u=u, r=self.payoff.reshape(n, 1, 1).copy(),
                    avail_u=one(n, 1, 2, 3), o_next=one(n, 1, 2, 1), s_next=one(n, 1, 1),
                    avail_u_next=one(n, 1, 2, 3), u_onehot=uo, padded=np.zeros((n, 1, 1)),
                    terminated=one(n, 1, 1))


def _params_t(agent):
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in agent.items()}


def serial_rollout(agent, args, env, n_episodes, epsilon, evaluate=False):
    """RolloutWorker.generate_episodes + SharedMAC.choose_action restated
    (rollout.py:30-173, controller/share_params.py:37-72).  Uses numpy's global RNG in the
    reference's draw order: one uniform per agent per step, one choice only when exploring.
    Returns (episodes dict float64/int64, rewards, wins, steps, epsilon_after)."""
    p = _params_t(agent)
    N, A, O, S, T = args.n_agents, args.n_actions, args.obs_shape, args.state_shape, args.episode_limit
    keys = ["o", "s", "u", "r", "avail_u", "o_next", "s_next", "avail_u_next", "u_onehot", "padded", "terminated"]
    out = {k: [] for k in keys}
    rewards, wins, steps_tot = [], [], 0
    self_eps = epsilon
    with torch.no_grad():
        for _ in range(n_episodes):
            env.reset()
            h = torch.zeros(N, args.rnn_hidden_dim)
            eps = 0 if evaluate else self_eps
            if args.epsilon_anneal_scale == "episode":
                eps = eps - args.anneal_epsilon if eps > args.min_epsilon else eps
            term, win, step, ep_r = False, False, 0, 0.0
            last = np.zeros((N, A))
            o, s, u, r, av, uo, te, pad = [], [], [], [], [], [], [], []
            while not term and step < T:
                obs, st, avail = env.get_obs(), env.get_state(), env.get_avail_actions()
                acts, onehots = [], []
                for i in range(N):
                    inp = np.hstack([obs[i]] + ([last[i]] if args.last_action else [])
                                    + ([np.eye(N)[i]] if args.reuse_network else []))
                    q, hi = nets.agent_step(p, torch.tensor(inp, dtype=torch.float32)[None], h[i:i + 1])
                    h[i] = hi[0]
                    q = q.clone()
                    q[torch.tensor(np.asarray(avail[i]), dtype=torch.float32)[None] == 0.0] = -float("inf")
                    if np.random.uniform() < eps:
                        a = int(np.random.choice(np.nonzero(avail[i])[0]))
                    else:
                        a = int(torch.argmax(q))
                    oh = np.zeros(A)
                    oh[a] = 1
                    acts.append(a)
                    onehots.append(oh)
                    last[i] = oh
                rew, term, info = env.step(acts)
                win = bool(term and info.get("battle_won", False))
                o.append(np.array(obs, dtype=np.float64)); s.append(np.array(st, dtype=np.float64))
                u.append(np.reshape(acts, [N, 1])); uo.append(np.array(onehots)); av.append(np.array(avail))
                r.append([rew]); te.append([float(term)]); pad.append([0.0])
                ep_r += rew
                step += 1
                if args.epsilon_anneal_scale == "step":
                    eps = eps - args.anneal_epsilon if eps > args.min_epsilon else eps
            o.append(np.array(env.get_obs(), dtype=np.float64)); s.append(np.array(env.get_state(), dtype=np.float64))
            av.append(np.array([env.get_avail_agent_actions(i) for i in range(N)]))
            o_n, s_n, av_n = o[1:], s[1:], av[1:]
            o, s, av = o[:-1], s[:-1], av[:-1]
            for _i in range(step, T):
                o.append(np.zeros((N, O))); u.append(np.zeros([N, 1])); s.append(np.zeros(S)); r.append([0.0])
                o_n.append(np.zeros((N, O))); s_n.append(np.zeros(S)); uo.append(np.zeros((N, A)))
                av.append(np.zeros((N, A))); av_n.append(np.zeros((N, A))); pad.append([1.0]); te.append([1.0])
            ep = dict(o=o, s=s, u=u, r=r, avail_u=av, o_next=o_n, s_next=s_n, avail_u_next=av_n,
                      u_onehot=uo, padded=pad, terminated=te)
            for k in keys:
                out[k].append(np.array(ep[k]))
            steps_tot += step
            wins.append(win)
            rewards.append(ep_r)
            if not evaluate:
                self_eps = eps
    episodes = {k: np.stack(v, 0) for k, v in out.items()}
    return episodes, rewards, wins, steps_tot, self_eps


def batched_rollout(agent, args, synth: SynthSMAC, n_envs, epsilon, evaluate=False, rseed=0,
                    env0=0, episode=0):
    """Lock-step rollout of ``n_envs`` SynthSMAC envs (ids env0..env0+n_envs-1).

    Per lock-step t (same epsilon for all envs, annealed once per lock-step - quirk Q7
    carried over): explore iff u01(key(rseed,EXPLORE,env,tg,n)) < eps, then the
    floor(u01(key(rseed,PICK,...)) * n_avail)-th available action, else first-index argmax
    of the avail-masked Q (share_params.py:66-70).  Finished envs emit padding
    (rollout.py:122-133).  Arrays are float32 / int64; layout as the reference."""
    p = _params_t(agent)
    N, A, O, S, T, H = args.n_agents, args.n_actions, args.obs_shape, args.state_shape, args.episode_limit, args.rnn_hidden_dim
    E = n_envs
    env = np.arange(env0, env0 + E)
    ep = np.full(E, episode)
    L = synth.length(env, ep)
    obs = np.zeros((E, T + 1, N, O), np.float32); st = np.zeros((E, T + 1, S), np.float32)
    av = np.zeros((E, T + 1, N, A), np.float32)
    u = np.zeros((E, T, N, 1), np.int64); uo = np.zeros((E, T, N, A), np.float32)
    r = np.zeros((E, T, 1), np.float32); term = np.ones((E, T, 1), np.float32); pad = np.ones((E, T, 1), np.float32)
    h = torch.zeros(E * N, H)
    last = np.zeros((E, N, A), np.float32)
    eps = 0.0 if evaluate else epsilon
    if args.epsilon_anneal_scale == "episode":
        eps = eps - args.anneal_epsilon if eps > args.min_epsilon else eps
    eye = np.eye(N, dtype=np.float32)
    with torch.no_grad():
        for t in range(T):
            alive = t < L
            if not alive.any():
                break
            o_t, s_t, a_t = synth.obs(env, ep, t), synth.state(env, ep, t), synth.avail(env, ep, t)
            parts = [o_t] + ([last] if args.last_action else []) + ([np.broadcast_to(eye, (E, N, N))] if args.reuse_network else [])
            inp = np.concatenate(parts, axis=-1).reshape(E * N, -1)
            q, h = nets.agent_step(p, torch.tensor(inp), h)
            q = q.numpy().reshape(E, N, A).copy()
            q[a_t == 0] = -np.inf
            greedy = q.argmax(-1)
            tg = synth.tg(ep, t)[:, None]
            explore = u01(key(rseed, ST_EXPLORE, env[:, None], tg, np.arange(N)[None])) < np.float32(eps)
            navail = a_t.sum(-1).astype(np.int64)
            k = np.floor(u01(key(rseed, ST_PICK, env[:, None], tg, np.arange(N)[None])) * navail.astype(np.float32)).astype(np.int64)
            k = np.minimum(k, navail - 1)
            csum = np.cumsum(a_t, -1)
            pick = (csum <= k[..., None]).sum(-1)
            act = np.where(explore, pick, greedy)
            rew = synth.reward(env, ep, t, act)
            oh = np.eye(A, dtype=np.float32)[act]
            m = alive
            obs[m, t], st[m, t], av[m, t] = o_t[m], s_t[m], a_t[m]
            u[m, t, :, 0] = act[m]; uo[m, t] = oh[m]; r[m, t, 0] = rew[m]
            pad[m, t, 0] = 0.0
            term[m, t, 0] = (t + 1 >= L[m]).astype(np.float32)
            last = oh
            fin = (t + 1 == L)
            if fin.any():   # last obs of an episode = o_next / avail_u_next of its final step
                obs[fin, t + 1] = synth.obs(env, ep, t + 1)[fin]
                st[fin, t + 1] = synth.state(env, ep, t + 1)[fin]
                av[fin, t + 1] = synth.avail(env, ep, t + 1)[fin]
            if args.epsilon_anneal_scale == "step":
                eps = eps - args.anneal_epsilon if eps > args.min_epsilon else eps
    o, o_n = obs[:, :-1].copy(), obs[:, 1:].copy()
    s, s_n = st[:, :-1].copy(), st[:, 1:].copy()
    a, a_n = av[:, :-1].copy(), av[:, 1:].copy()
    for e in range(E):   # padding rows are zero in every key (rollout.py:122-133)
        o_n[e, L[e]:] = 0; s_n[e, L[e]:] = 0; a_n[e, L[e]:] = 0
        o[e, L[e]:] = 0; s[e, L[e]:] = 0; a[e, L[e]:] = 0
    episodes = dict(o=o, s=s, u=u, r=r, avail_u=a, o_next=o_n, s_next=s_n, avail_u_next=a_n,
                    u_onehot=uo, padded=pad, terminated=term)
    rewards = [float(r[e, :, 0].sum()) for e in range(E)]
    wins = list(synth.won(env, ep))
    return episodes, rewards, wins, int(L.sum()), (epsilon if evaluate else eps)
