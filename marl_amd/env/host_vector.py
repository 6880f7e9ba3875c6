"""HostVectorEnv: n HOST environments that speak the reference's serial env API, driven in lock-step.

The reference steps ONE environment and ONE agent at a time (rollout.py:42 ``env.reset()``, :61-64 ``get_obs /
get_state / get_avail_actions``, :86-88 ``env.step(actions)``; main.py:16-29 builds the env and reads
``get_env_info()``).  A real SMAC install gives exactly such objects.  This adapter puts n of them behind the
batched protocol RolloutWorker drives (``new_record / begin_episode / observe / step``, the protocol of
env/synthetic_smac.py), so the agent step and the epsilon-greedy choice of ALL n environments are one
``agent_unroll_fwd(T=1)`` + one ``select_actions`` launch per lock-step instead of n x N launches:

    per lock-step:  ONE D2H copy of the chosen actions (E x N int32, pinned)
                    n x env.step() on the host, then get_obs / get_state / get_avail_actions of the envs still running
                    ONE H2D copy of the bundle [obs | state | avail | r | term | padded | alive | length | won] (pinned)
                    a handful of strided device copies that scatter the bundle into slot t+1 / step t of the record

Everything numeric stays where it was: the environments compute on the host (they are the user's), the network on the
HIP kernels.  The record obeys the reference's padding rules (rollout.py:122-133): rows of finished episodes are zero,
``padded`` = ``terminated`` = 1, ``u`` = -1 (EpisodeBatch turns that into the reference's 0 / zero one-hot).
Random draws: the batched path's epsilon-greedy stream is the counter hash of csrc/rollout.hip (env id = position in
``envs`` + ``env0``), not numpy's global stream - see DESIGN section 2.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..hostutil import require_cuda
from .synthetic_smac import EpisodeRecord


class HostVectorEnv:
    batched = True

    def __init__(self, envs, seed=1, env0=0, n_threads=0):
        """``envs``: a list of environment objects with the reference's API, or a zero-argument factory + count as
        ``(factory, n)``.  ``n_threads`` > 0 steps / observes the environments on a thread pool (SMAC environments block
        on a socket to their StarCraft II process; the calls of different environments are independent)."""
        if isinstance(envs, tuple) and callable(envs[0]):
            envs = [envs[0]() for _ in range(int(envs[1]))]
        self.envs = list(envs)
        if not self.envs:
            raise ValueError("HostVectorEnv needs at least one environment")
        self.n_envs = len(self.envs)
        info = self.envs[0].get_env_info()
        self.n_actions, self.n_agents = int(info["n_actions"]), int(info["n_agents"])
        self.state_shape, self.obs_shape = int(info["state_shape"]), int(info["obs_shape"])
        self.episode_limit = int(info["episode_limit"])
        self.seed, self.env0 = seed, env0
        self.episode = -1
        self.device = require_cuda("HostVectorEnv")
        N, O, S, A = self.n_agents, self.obs_shape, self.state_shape, self.n_actions
        # bundle row of one environment: [obs N*O | state S | avail N*A | r | term | padded | alive_next | length | won]
        self._o0, self._s0, self._a0 = 0, N * O, N * O + S
        self._x0 = N * O + S + N * A
        self.width = self._x0 + 6
        E = self.n_envs
        self._stage_h = torch.zeros(E, self.width, dtype=torch.float32).pin_memory()
        self._stage_np = self._stage_h.numpy()
        self._stage_d = torch.zeros(E, self.width, dtype=torch.float32, device=self.device)
        self._act_h = torch.zeros(E, N, dtype=torch.int32).pin_memory()
        self._act_np = self._act_h.numpy()
        self._act_ev = torch.cuda.Event()
        self._alive = np.zeros(E, dtype=bool)
        self._length = np.zeros(E, dtype=np.int64)
        self._won = np.zeros(E, dtype=bool)
        self._staged_slot = -1
        self._pool = ThreadPoolExecutor(n_threads) if n_threads and n_threads > 1 else None
        self.h2d_copies = self.d2h_copies = 0       # counted for the tests: one of each per lock-step

    # ------------------------------------------------------------------ reference env surface (main.py:22-29, :44)
    def get_env_info(self):
        return {"n_actions": self.n_actions, "n_agents": self.n_agents, "state_shape": self.state_shape,
                "obs_shape": self.obs_shape, "episode_limit": self.episode_limit}

    def close(self):
        for e in self.envs:
            e.close()

    def save_replay(self):
        for e in self.envs:
            e.save_replay()

    # ------------------------------------------------------------------ batched protocol
    def new_record(self):
        return EpisodeRecord(self.n_envs, self.episode_limit, self.n_agents, self.obs_shape, self.state_shape,
                             self.n_actions, self.device)

    def global_step(self, t):
        return self.episode * (self.episode_limit + 1) + t

    def _map(self, fn, idx):
        if self._pool is not None and len(idx) > 1:
            return list(self._pool.map(fn, idx))
        return [fn(i) for i in idx]

    def begin_episode(self, rec):
        self.episode += 1
        torch.cuda.current_stream().synchronize()      # the previous rollout's last bundle has left the pinned buffer
        self._map(lambda i: self.envs[i].reset(), range(self.n_envs))        # rollout.py:42
        self._alive[:] = True
        self._length[:] = 0
        self._won[:] = False
        self._stage_np[:] = 0.0
        self._staged_slot = -1

    def _read_obs(self, i, final=False):
        """slot of environment i into its bundle row (rollout.py:61-64; the slot after the last step: :104-113)"""
        env, row = self.envs[i], self._stage_np[i]
        N, O, A = self.n_agents, self.obs_shape, self.n_actions
        row[self._o0:self._s0] = np.asarray(env.get_obs(), dtype=np.float32).reshape(N * O)
        row[self._s0:self._a0] = np.asarray(env.get_state(), dtype=np.float32).reshape(self.state_shape)
        if final:
            av = [env.get_avail_agent_actions(n) for n in range(N)]
        else:
            av = env.get_avail_actions()
        row[self._a0:self._x0] = np.asarray(av, dtype=np.float32).reshape(N * A)

    def _upload(self, rec, slot, step):
        """the ONE host-to-device copy of a lock-step, then scatter the bundle into the record: slot ``slot`` of
        obs / state / avail, and (``step`` >= 0) r / term / padded of step ``step`` + the episode ends known so far"""
        E, N, O, S, A = self.n_envs, self.n_agents, self.obs_shape, self.state_shape, self.n_actions
        self._stage_d.copy_(self._stage_h, non_blocking=True)
        self.h2d_copies += 1
        d = self._stage_d
        rec.obs[:, slot].copy_(d[:, self._o0:self._s0].view(E, N, O))
        rec.state[:, slot].copy_(d[:, self._s0:self._a0])
        rec.avail[:, slot].copy_(d[:, self._a0:self._x0].view(E, N, A))
        if step >= 0:
            x = self._x0
            rec.r[:, step].copy_(d[:, x])
            rec.term[:, step].copy_(d[:, x + 1])
            rec.padded[:, step].copy_(d[:, x + 2])
            rec.length.copy_(d[:, x + 4])
            rec.won.copy_(d[:, x + 5])
        self._staged_slot = slot
        # the pinned bundle is overwritten by the next lock-step's host work: that work starts only after the D2H of the
        # next actions, which is ordered after this copy on the same stream - no extra synchronisation needed

    def observe(self, t, rec):
        if self._staged_slot == t:        # step(t-1) already shipped slot t with its results
            return
        self._map(self._read_obs, [i for i in range(self.n_envs) if self._alive[i]])
        self._upload(rec, t, -1)

    def step(self, t, act, rec, alive_next):
        """``act`` (E, N) int32 on the device, ``alive_next`` (E) int32 on the device (1 while an episode runs)"""
        E, x = self.n_envs, self._x0
        rec.u[:, t].copy_(act)          # select_actions wrote -1 (padding) for the episodes that are over
        self._act_h.copy_(act, non_blocking=True)                          # the ONE device-to-host copy of a lock-step
        self._act_ev.record()
        self.d2h_copies += 1
        self._act_ev.synchronize()
        st = self._stage_np
        was_alive = self._alive.copy()
        running = [i for i in range(E) if was_alive[i]]

        def one(i):
            reward, terminated, info = self.envs[i].step([int(a) for a in self._act_np[i]])   # rollout.py:86
            done = bool(terminated) or t + 1 >= self.episode_limit
            self._length[i] = t + 1
            if bool(terminated) and isinstance(info, dict) and info.get('battle_won'):       # rollout.py:87
                self._won[i] = True
            st[i, x:x + 3] = (reward, 1.0 if terminated else 0.0, 0.0)
            self._alive[i] = not done
            self._read_obs(i, final=done)          # o_next of the last step is the observation after it (rollout.py:104)
            return done

        self._map(one, running)
        dead = ~was_alive
        if dead.any():                                 # padding rows (rollout.py:122-133)
            st[dead, :x] = 0.0
            st[dead, x:x + 3] = (0.0, 1.0, 1.0)
        st[:, x + 3] = self._alive
        st[:, x + 4] = self._length
        st[:, x + 5] = self._won
        self._upload(rec, t + 1, t)
        alive_next.copy_(self._stage_d[:, x + 3])
