"""Two-agent single-state matrix game (mirror of reference env/single_state_matrix_game.py:5-120)
plus a vectorised variant for lock-step rollouts (BASELINE config 1: 32 parallel envs)."""
from __future__ import annotations

import datetime

import numpy as np
import torch

from .synthetic_smac import EpisodeRecord
from ..hostutil import require_cuda


class TwoAgentsMatrixGame:
    def __init__(self, payoff_table, replay_dir='./replay_dir'):
        self.payoff_table = np.array(payoff_table, dtype=float)
        self._init_replay()
        self.current_episode = 0
        self.replay_dir = replay_dir
        self.n_actions, self.n_agents, self.state_shape, self.obs_shape, self.episode_limit = 3, 2, 1, 1, 1
        self.env_info = {"n_actions": 3, "n_agents": 2, "state_shape": 1, "obs_shape": 1, "episode_limit": 1,
                         "env_name": "SingleStateMatrixGame"}

    def step(self, actions):
        reward = self.payoff_table[int(actions[0]), int(actions[1])]
        rep = self.replay[self.current_episode]
        rep["obs"].append([1., 1.]); rep["state"].append(1.); rep["actions"].append(actions)
        rep["reward"].append(reward); rep["episode_length"] += 1
        return reward, True, {}

    def get_obs(self):          # quirk Q9: zeros during rollout ...
        return [np.array([0.]), np.array([0.])]

    def get_state(self):
        return np.array([0.])

    def get_avail_actions(self):
        return [np.array([1, 1, 1]), np.array([1, 1, 1])]

    def get_avail_agent_actions(self, agent_id):
        return np.array([1, 1, 1])

    def reset(self):
        if self.replay[self.current_episode]["episode_length"] != 0:
            self.replay.append({"obs": [], "state": [], "actions": [], "reward": [], "episode_length": 0})
            self.current_episode += 1

    def close(self):
        self._init_replay()
        self.current_episode = 0

    def _init_replay(self):
        self.replay = [{"obs": [], "state": [], "actions": [], "reward": [], "episode_length": 0}]

    def save_replay(self):
        stamp = datetime.datetime.today().strftime('%Y-%m-%d_%H %M %S')
        np.save(self.env_info["env_name"] + stamp, np.array(self.replay, dtype=object), allow_pickle=True)

    def get_env_info(self):
        return self.env_info

    def get_episodes(self):     # ... ones in the fixed 9-episode training batch (reference :81-120)
        n = self.payoff_table.size
        u = np.zeros((n, 1, 2, 1), dtype=np.int64)
        uo = np.zeros((n, 1, 2, 3), dtype=np.int64)
        for i in range(n):
            a0, a1 = divmod(i, 3)        # cartesian order (a0,a1) = (0,0),(0,1),(0,2),(1,0)...
            u[i, 0, :, 0] = (a0, a1)
            uo[i, 0, 0, a0] = 1
            uo[i, 0, 1, a1] = 1
        ones = lambda *s: np.ones(s)
        return dict(o=ones(n, 1, 2, 1), s=ones(n, 1, 1), u=u, r=self.payoff_table.reshape(n, 1, 1).copy(),
                    avail_u=ones(n, 1, 2, 3), o_next=ones(n, 1, 2, 1), s_next=ones(n, 1, 1),
                    avail_u_next=ones(n, 1, 2, 3), u_onehot=uo, padded=np.zeros((n, 1, 1)), terminated=ones(n, 1, 1))


class BatchedMatrixGame:
    """n_envs copies of the game stepped in lock-step on the device (observations are zeros as in
    the serial rollout; one step per episode)."""
    batched = True

    def __init__(self, payoff_table, n_envs, seed=1):
        self.payoff = np.array(payoff_table, dtype=np.float32)
        self.n_envs = n_envs
        self.n_actions, self.n_agents, self.state_shape, self.obs_shape, self.episode_limit = 3, 2, 1, 1, 1
        self.device = require_cuda("BatchedMatrixGame")
        self.payoff_d = torch.tensor(self.payoff, device=self.device)
        self.episode = -1
        self.seed, self.env0 = seed, 0

    def get_env_info(self):
        return {"n_actions": 3, "n_agents": 2, "state_shape": 1, "obs_shape": 1, "episode_limit": 1}

    def new_record(self):
        return EpisodeRecord(self.n_envs, 1, 2, 1, 1, 3, self.device)

    def begin_episode(self, rec):
        self.episode += 1
        rec.length.fill_(1)
        rec.won.zero_()

    def observe(self, t, rec):
        rec.obs[:, t].zero_()
        rec.state[:, t].zero_()
        rec.avail[:, t].fill_(1.0)

    def step(self, t, act, rec, alive_next):
        a = act.long()
        rec.u[:, t] = act
        rec.r[:, t] = self.payoff_d[a[:, 0], a[:, 1]]     # table lookup: data movement, no arithmetic
        rec.term[:, t] = 1.0
        rec.padded[:, t] = 0.0
        alive_next.zero_()

    def global_step(self, t):
        return self.episode * 2 + t

    def close(self):
        pass

    def save_replay(self):
        pass
