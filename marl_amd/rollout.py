"""RolloutWorker (mirror of reference rollout.py:3-173).

* env with the serial SMAC API  -> the reference's serial loop (one env, one agent at a time,
  numpy global RNG in the reference's draw order), network calls on the HIP kernel.
* env with ``batched = True``   -> all ``env.n_envs`` environments advance in lock-step on the
  device: per step one observe kernel, one agent-step kernel (the T=1 unroll), one epsilon-greedy
  selection kernel, one env-step kernel; the episode record never leaves HBM.
  Epsilon is annealed once per lock-step (quirk Q7 kept for n_envs = 1).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .hostutil import h2d_async, require_cuda


class EpisodeBatch(dict):
    """The 11-key episode dict of the reference (rollout.py:135-146) materialised lazily from a
    device EpisodeRecord; learners / ReplayBuffer take ``.record`` directly (no copies)."""

    KEYS = ("o", "s", "u", "r", "avail_u", "o_next", "s_next", "avail_u_next", "u_onehot", "padded", "terminated")

    def __init__(self, record=None, ring=None, index=None):
        """``record``: the episodes themselves; or ``ring`` + ``index``: a replay sample that learners read in
        place (the gathered copy is only made if somebody asks for ``.record`` / a dict key)."""
        super().__init__()
        self._record = record
        self.ring, self.index = ring, index

    @property
    def record(self):
        if self._record is None and self.ring is not None:
            self._record = self.ring.index_select(self.index)
        return self._record

    def _build(self, k):
        r = self.record
        T = r.T
        t_idx = torch.arange(T, device=r.obs.device)[None, :]
        live = (t_idx < r.length[:, None])               # (E,T) real steps
        if k in ("o", "s", "avail_u"):
            src = {"o": r.obs, "s": r.state, "avail_u": r.avail}[k][:, :T]
            m = live.view(r.E, T, *([1] * (src.dim() - 2)))
            return torch.where(m, src, torch.zeros_like(src))    # padded rows are zero (rollout.py:122-133)
        if k in ("o_next", "s_next", "avail_u_next"):
            return {"o_next": r.obs, "s_next": r.state, "avail_u_next": r.avail}[k][:, 1:]
        if k == "u":
            return r.u.clamp(min=0).long().unsqueeze(-1)
        if k == "u_onehot":
            oh = torch.zeros(r.E, T, r.N, r.A, device=r.u.device)
            return oh.scatter_(3, r.u.clamp(min=0).long().unsqueeze(-1), (r.u >= 0).float().unsqueeze(-1))
        if k == "u_idx":
            return r.u
        return {"r": r.r, "padded": r.padded, "terminated": r.term}[k].unsqueeze(-1)

    def __missing__(self, k):
        if k not in self.KEYS and k != "u_idx":
            raise KeyError(k)
        v = self._build(k)
        self[k] = v
        return v

    def keys(self):
        return list(self.KEYS)

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)

    def items(self):
        return [(k, self[k]) for k in self.KEYS]

    def numpy(self):
        """Host copy in the reference's dtypes (float64 / int64)."""
        out = {}
        for k in self.KEYS:
            v = self[k].cpu().numpy()
            out[k] = v.astype(np.int64) if k == "u" else v.astype(np.float64)
        return out


class RolloutStats:
    """Per-episode reward / win flag / length of one rollout, copied to pinned host memory in stream order; the
    accessors wait for that copy only (finish_episodes(lazy=True))."""

    _pool = {}        # shape -> free pinned buffers (a fresh pin_memory() per rollout is a host allocation call)

    def __init__(self, stats_dev):
        self._key = (tuple(stats_dev.shape), stats_dev.dtype)
        free = RolloutStats._pool.setdefault(self._key, [])
        self.buf = free.pop() if free else torch.empty(stats_dev.shape, dtype=stats_dev.dtype).pin_memory()
        self.buf.copy_(stats_dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def _host(self):
        self.event.synchronize()
        return self.buf

    def __del__(self):            # the buffer goes back to the pool with its owner (nobody reads it any more)
        try:
            RolloutStats._pool[self._key].append(self.buf)
        except Exception:
            pass

    def rewards(self):
        return self._host()[0].tolist()

    def wins(self):
        return [bool(x) for x in self._host()[1].tolist()]

    def steps(self):
        return int(self._host()[2].sum().item())


class RolloutWorker:
    def __init__(self, env, mac, args):
        self.env = env
        self.mac = mac
        self.episode_limit = args.episode_limit
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.state_shape = args.state_shape
        self.obs_shape = args.obs_shape
        self.args = args
        self.epsilon = args.epsilon
        self.anneal_epsilon = args.anneal_epsilon
        self.min_epsilon = args.min_epsilon
        self.rseed = getattr(args, "seed", 0)
        self._bufs = {}
        print('Init RolloutWorker')

    def init_last_actions(self):
        return np.zeros((self.args.n_agents, self.args.n_actions))

    # ------------------------------------------------------------------ public API (reference :30)
    def generate_episodes(self, n_episodes=1, evaluate=False, random_select=False):
        if getattr(self.env, "batched", False):
            return self._generate_batched(evaluate)
        return self._generate_serial(n_episodes, evaluate, random_select)

    # ------------------------------------------------------------------ batched device path
    def _generate_batched(self, evaluate):
        env, mac, a = self.env, self.mac, self.args
        dev = require_cuda("RolloutWorker")
        E, T, N, A, O, H = env.n_envs, self.episode_limit, self.n_agents, self.n_actions, self.obs_shape, a.rnn_hidden_dim
        if a.replay_dir != '' and evaluate:
            env.close()
        mode = getattr(self, "rollout_mode", "whole")      # "whole" | "fused_step" | "unfused" (tests)
        if mode == "whole" and hasattr(env, "whole_rollout") and env.supports_whole_rollout():
            # the persistent kernel writes every field of the record, so training rollouts can be
            # played straight into the replay ring (record_sink = the ReplayBuffer; zero-copy store)
            sink = getattr(self, "record_sink", None)
            rec = None
            if sink is not None and not evaluate:
                rec = sink.next_slot_record(E, T, N, O, self.state_shape, A, dev)
            if rec is None:
                rec = env.new_record()
            return self._generate_whole(rec, evaluate)
        rec = env.new_record()
        env.begin_episode(rec)
        mac.init_hidden(E)
        h = mac.hidden_states.view(E * N, H)
        b = self._bufs
        if b.get("E") != E:
            b.update(E=E, q=torch.empty(E, 1, N, A, device=dev), act=torch.empty(E, N, dtype=torch.int32, device=dev),
                     alive=torch.empty(E, dtype=torch.int32, device=dev))
        q, act, alive = b["q"], b["act"], b["alive"]
        alive.fill_(1)
        epsilon = 0 if evaluate else self.epsilon
        if a.epsilon_anneal_scale == 'episode':
            epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
        w = mac.agent.weights()
        fused = hasattr(env, "fused_step") and mode != "unfused"
        env.observe(0, rec)
        for t in range(T):
            # agent step = the unroll kernel with T=1 reading slot t of the record in place
            ops.agent_unroll_fwd(w, rec.obs, (T + 1) * N, t, rec.u, T * N, t - 1, h, q, None, h, None,
                                 E, 1, N, O, A, a.last_action, a.reuse_network)
            if fused:
                env.fused_step(t, q, epsilon, self.rseed, rec)
            else:
                ops.select_actions(q, rec.avail[:, t], (T + 1) * N * A, alive, epsilon, self.rseed, env.env0, None,
                                   env.global_step(t), act, N, E, N, A)
                env.step(t, act, rec, alive)
                env.observe(t + 1, rec)
            if a.epsilon_anneal_scale == 'step':
                epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
        if not evaluate:
            self.epsilon = epsilon
        if evaluate and a.replay_dir != '':
            env.save_replay()
            env.close()
        stats = torch.stack([rec.r.sum(1), rec.won.float() , rec.length.float()], 0).cpu()   # one D2H copy
        episodes_reward = stats[0].tolist()
        wins_tag = [bool(x) for x in stats[1].tolist()]
        steps_tot = int(stats[2].sum().item())
        return EpisodeBatch(rec), episodes_reward, wins_tag, steps_tot

    def _generate_whole(self, rec, evaluate):
        return self.finish_episodes(self._launch_whole(rec, evaluate, self.mac))

    def _launch_whole(self, rec, evaluate, mac):
        """One persistent launch for the whole rollout; the epsilon schedule (one anneal per lock-step,
        reference rollout.py:48-50,100-101) is evaluated on the host and shipped as a T-vector.  Everything is
        enqueued on the current HIP stream; nothing here waits for the GPU."""
        env, a = self.env, self.args
        dev = require_cuda("RolloutWorker")
        E, T, N, H = env.n_envs, self.episode_limit, self.n_agents, a.rnn_hidden_dim
        if a.replay_dir != '' and evaluate:
            env.close()
        epsilon = 0 if evaluate else self.epsilon
        if a.epsilon_anneal_scale == 'episode':
            epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
        # the kernel evaluates the same fp64 recurrence per lock-step (no schedule vector crosses PCIe); the host runs it
        # only to know the epsilon the next rollout starts from
        step_scale = a.epsilon_anneal_scale == 'step'
        sched = (float(epsilon), float(self.anneal_epsilon) if step_scale else 0.0, float(self.min_epsilon))
        if step_scale:
            for t in range(T):
                epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
        mac.init_hidden(E)
        from .network import mixer as _mixer
        x6 = getattr(a, "gemm_mode", _mixer.DEFAULT_GEMM_MODE) == "bf16x6"      # the agent step as split products (csrc/rollout_x6.hip)
        env.whole_rollout(mac.agent.weights(), None, self.rseed, rec, a.last_action, a.reuse_network,
                          h_out=mac.hidden_states.view(E * N, H), eps_sched=sched, x6=x6)
        if not evaluate:
            self.epsilon = epsilon
        return rec, evaluate

    def launch_episodes(self, evaluate=False, mac=None):
        """Asynchronous form of generate_episodes for batched envs with the whole-rollout kernel: enqueue the rollout
        on the CURRENT stream (a side stream in the overlapped runner) and return a handle for finish_episodes().
        ``mac``: the controller whose weights the rollout reads (a snapshot while the learner updates the live one)."""
        env = self.env
        if not (getattr(env, "batched", False) and hasattr(env, "whole_rollout") and env.supports_whole_rollout()):
            raise RuntimeError("launch_episodes needs a batched env with the whole-rollout kernel")
        dev = require_cuda("RolloutWorker")
        sink = getattr(self, "record_sink", None)
        rec = None
        if sink is not None and not evaluate:
            rec = sink.next_slot_record(env.n_envs, self.episode_limit, self.n_agents, self.obs_shape, self.state_shape,
                                        self.n_actions, dev)
        if rec is None:
            rec = env.new_record()
        return self._launch_whole(rec, evaluate, mac if mac is not None else self.mac)

    def finish_episodes(self, pending, lazy=False):
        """(episodes, rewards, win_tags, steps) of a launched rollout: the one device-to-host copy (and sync).
        ``lazy``: the statistics are still reduced on the device and copied out, but into pinned memory in stream order;
        returns (episodes, RolloutStats) and the host does not wait - RolloutStats.rewards() / wins() / steps() do."""
        rec, evaluate = pending
        if evaluate and self.args.replay_dir != '':
            self.env.save_replay()
            self.env.close()
        stats = getattr(rec, "kernel_stats", None)         # written by the whole-rollout kernel (one launch, no reduction pass)
        if stats is not None:
            rec.kernel_stats = None
        else:
            stats = torch.stack([rec.r.sum(1), rec.won.float(), rec.length.float()], 0)
        if lazy:
            return EpisodeBatch(rec), RolloutStats(stats)
        stats = stats.cpu()
        return EpisodeBatch(rec), stats[0].tolist(), [bool(x) for x in stats[1].tolist()], int(stats[2].sum().item())

    # ------------------------------------------------------------------ serial path (reference loop)
    def _generate_serial(self, n_episodes, evaluate, random_select):
        a = self.args
        N, A = self.n_agents, self.n_actions
        steps_tot, wins_tag, episodes_reward = 0, [], []
        if a.replay_dir != '' and evaluate:
            self.env.close()
        keys = EpisodeBatch.KEYS
        collected = {k: [] for k in keys}
        for num_episode in range(n_episodes):
            self.env.reset()
            self.mac.init_hidden(1)
            epsilon = 0 if evaluate else self.epsilon
            if a.epsilon_anneal_scale == 'episode':
                epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
            terminated, win_tag, step, episode_reward = False, False, 0, 0
            last_actions = self.init_last_actions()
            o, u, r, s, avail_u, u_onehot, terminate, padded = [], [], [], [], [], [], [], []
            while not terminated and step < self.episode_limit:
                obs = self.env.get_obs()
                state = self.env.get_state()
                avail_actions = self.env.get_avail_actions()
                actions, actions_onehot = [], []
                for agent_id in range(N):
                    if random_select:   # quirk Q8: randint(0, A-1) never draws the last action
                        action = np.random.randint(0, A - 1)
                        while avail_actions[agent_id][action] == 0:
                            action = np.random.randint(0, A - 1)
                    else:
                        action = self.mac.choose_action(obs[agent_id], last_actions[agent_id], agent_id,
                                                        avail_actions[agent_id], epsilon, evaluate)
                    onehot = np.zeros(A)
                    onehot[action] = 1
                    actions.append(action)
                    actions_onehot.append(onehot)
                    last_actions[agent_id] = onehot
                reward, terminated, info = self.env.step(actions)
                win_tag = True if terminated and 'battle_won' in info and info['battle_won'] else False
                o.append(obs); s.append(state)
                u.append(np.reshape([int(x) for x in actions], [N, 1]))
                u_onehot.append(actions_onehot); avail_u.append(avail_actions)
                r.append([reward]); terminate.append([terminated]); padded.append([0.])
                episode_reward += reward
                step += 1
                if a.epsilon_anneal_scale == 'step':
                    epsilon = epsilon - self.anneal_epsilon if epsilon > self.min_epsilon else epsilon
            o.append(self.env.get_obs()); s.append(self.env.get_state())
            o_next, s_next, o, s = o[1:], s[1:], o[:-1], s[:-1]
            avail_u.append([self.env.get_avail_agent_actions(i) for i in range(N)])
            avail_u_next, avail_u = avail_u[1:], avail_u[:-1]
            for _ in range(step, self.episode_limit):
                o.append(np.zeros((N, self.obs_shape))); u.append(np.zeros([N, 1])); s.append(np.zeros(self.state_shape))
                r.append([0.]); o_next.append(np.zeros((N, self.obs_shape))); s_next.append(np.zeros(self.state_shape))
                u_onehot.append(np.zeros((N, A))); avail_u.append(np.zeros((N, A))); avail_u_next.append(np.zeros((N, A)))
                padded.append([1.]); terminate.append([1.])
            ep = dict(o=o, s=s, u=u, r=r, avail_u=avail_u, o_next=o_next, s_next=s_next, avail_u_next=avail_u_next,
                      u_onehot=u_onehot, padded=padded, terminated=terminate)
            for k in keys:
                collected[k].append(np.array(ep[k]))
            steps_tot += step
            wins_tag.append(win_tag)
            episodes_reward.append(episode_reward)
            if evaluate and num_episode == n_episodes - 1 and a.replay_dir != '':
                self.env.save_replay()
                self.env.close()
            if not evaluate:
                self.epsilon = epsilon
        episodes = {k: np.stack(v, axis=0) for k, v in collected.items()} if n_episodes > 0 else None
        return episodes, episodes_reward, wins_tag, steps_tot
