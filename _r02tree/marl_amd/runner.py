"""Runner (mirror of reference runner.py:14-141): wires MAC + RolloutWorker + ReplayBuffer + learner
and reproduces the train / evaluate cadence, the logging tags and the save cycle.  With a batched env
every iteration collects ``env.n_envs`` episodes in lock-step and trains on ``batch_size`` sampled
episodes; with a serial env it behaves exactly like the reference loop.

``args.overlap_rollout`` (opt-in, batched envs): the rollout of iteration k+1 runs on a side HIP stream while the
learner trains on iteration k's sample.  The rollout reads a SNAPSHOT of the agent taken before update k (policy lag
of one update - the reference has lag 0, which is why this is off by default), writes into ring slots the sampler
excludes, and the cadence of runner.py:85-98 (store -> train_steps updates -> log -> save) is unchanged.
``"lag1_serial"`` runs the same schedule on one stream (the parity check of the overlapped mode: identical losses).

Full resume: ``save_resume`` / ``load_resume`` carry what the reference's checkpoints lack - optimizer
state, target networks, epsilon, the loop counters, the numpy RNG state (the replay ring refills)."""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from .rollout import RolloutWorker
from .controller.share_params import SharedMAC
from .common.replaybuffer import ReplayBuffer
from .algorithm.q_learner import QLearner
from .algorithm.qtran_learner import QTRANLearner
from .utils.logging import Logger


class Runner:
    def __init__(self, env, logger, args):
        self.env = env
        if not args.reuse_network or getattr(args, "RTW", False):
            raise NotImplementedError("only the shared-parameter controller (reuse_network, RTW off) is on the hot path")
        self.mac = SharedMAC(args)
        self.rolloutWorker = RolloutWorker(env, self.mac, args)
        self.buffer = ReplayBuffer(args)
        self.rolloutWorker.record_sink = self.buffer   # batched rollouts write into the replay ring in place
        self.args = args
        self.eval_win_rates = []
        self.eval_episode_rewards = []
        if self.args.env in ('smac', 'synthetic', 'matrix'):
            self.save_path = self.args.result_dir + '/' + args.alg + '/' + args.map
        else:
            raise ValueError("env {} dose not exist!".format(self.args.env))
        os.makedirs(self.save_path, exist_ok=True)
        logger.setup_tb(self.save_path + '/tb/other')
        self.logger = logger
        if any(args.alg.find(a) > -1 for a in ('vdn', 'qmix', 'qplex')):
            self.learner = QLearner(self.mac, args)
        elif args.alg.find('qtran_base') > -1 or args.alg.find('qtran_alt') > -1:
            self.learner = QTRANLearner(self.mac, args)
        else:
            raise ValueError('learner {} cannot find!'.format(args.alg))
        if args.load_model:
            self.learner.load_models()
        self.time_steps, self.train_steps, self.evaluate_steps = 0, 0, -1     # loop counters (restored by load_resume)
        self.losses = []
        self.overlap = getattr(args, "overlap_rollout", False)
        self._side = None
        self._mac_roll = None
        if getattr(args, "resume", ""):
            self.load_resume(args.resume)

    # ------------------------------------------------------------------ overlapped rollout
    def _launch_rollout(self):
        """snapshot the agent, then enqueue the next training rollout - on the side stream when overlapping"""
        from .hostutil import flatten_module
        if self._mac_roll is None:
            self._mac_roll = copy.deepcopy(self.mac)
            flatten_module(self._mac_roll.agent, self.learner.device)
            self._mac_roll._dev = self.learner.device
        cur = torch.cuda.current_stream()
        if self.overlap is True:
            if self._side is None:
                self._side = torch.cuda.Stream(device=cur.device)
            self._side.wait_stream(cur)         # the snapshot reads what the main stream has written so far
            ctx = torch.cuda.stream(self._side)
        else:
            ctx = torch.cuda.stream(cur)
        with ctx:
            self._mac_roll.agent._flat.flat.copy_(self.mac.agent._flat.flat)
            pending = self.rolloutWorker.launch_episodes(mac=self._mac_roll)
        slot = getattr(pending[0], "sink_slot", None)
        return pending, (None if slot is None else (slot, pending[0].E))

    def _finish_rollout(self, pending):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        return self.rolloutWorker.finish_episodes(pending)

    def run(self, num):
        """reference runner.py:61-113."""
        a = self.args
        n_ep = getattr(self.env, "n_envs", a.n_episodes)
        loss = float("nan")
        pending, in_flight = None, None
        while self.time_steps < a.n_steps:
            if self.time_steps // a.evaluate_cycle > self.evaluate_steps:
                win_rate, episode_reward = self.evaluate()
                self.eval_win_rates.append(win_rate)
                self.eval_episode_rewards.append(episode_reward)
                self.plt(num)
                self.logger.log_stat("test_win_rate", win_rate, self.time_steps)
                self.logger.log_stat("test_episode_reward", episode_reward, self.time_steps)
                self.evaluate_steps += 1
            if self.overlap:
                if pending is None:
                    pending, _ = self._launch_rollout()          # first iteration: nothing to overlap with yet
                episodes, rewards, win_tags, steps = self._finish_rollout(pending)
            else:
                episodes, rewards, win_tags, steps = self.rolloutWorker.generate_episodes(n_episodes=n_ep, random_select=False)
            self.time_steps += steps
            self.logger.log_stat("episode_length", steps, self.time_steps)
            self.logger.log_stat("train_win_rate", sum(win_tags) / n_ep, self.time_steps)
            self.logger.log_stat("train_episode_reward", sum(rewards) / n_ep, self.time_steps)
            self.buffer.store_episode(episodes)
            if self.overlap:
                pending, in_flight = self._launch_rollout()      # rollout k+1 flies while update k trains below
            for _ in range(a.train_steps):
                mini_batch = self.buffer.sample(min(self.buffer.current_size, a.batch_size), exclude=in_flight)
                loss = self.learner.train(mini_batch, self.train_steps)
                self.losses.append(loss)
                self.train_steps += 1
            self.logger.log_stat("total_loss", loss, self.time_steps)
            if self.train_steps > 0 and self.train_steps % a.save_cycle == 0:
                self.learner.save_models(self.train_steps)
                if getattr(a, "save_resume", True):
                    self.save_resume(self.save_path + '/resume.pt')
        if pending is not None:
            self._finish_rollout(pending)                        # drain the rollout still in flight
        win_rate, episode_reward = self.evaluate()
        self.eval_win_rates.append(win_rate)
        self.eval_episode_rewards.append(episode_reward)
        self.plt(num)
        return loss

    # ------------------------------------------------------------------ full resume
    def save_resume(self, path):
        torch.save({"learner": self.learner.resume_state(), "epsilon": self.rolloutWorker.epsilon,
                    "time_steps": self.time_steps, "train_steps": self.train_steps, "evaluate_steps": self.evaluate_steps,
                    "eval_win_rates": list(self.eval_win_rates), "eval_episode_rewards": list(self.eval_episode_rewards),
                    "env_episode": getattr(self.env, "episode", None), "np_random": np.random.get_state()}, path)

    def load_resume(self, path):
        sd = torch.load(path, map_location="cpu", weights_only=False)
        self.learner.load_resume_state(sd["learner"])
        self.rolloutWorker.epsilon = sd["epsilon"]
        self.time_steps, self.train_steps, self.evaluate_steps = sd["time_steps"], sd["train_steps"], sd["evaluate_steps"]
        self.eval_win_rates, self.eval_episode_rewards = list(sd["eval_win_rates"]), list(sd["eval_episode_rewards"])
        if sd.get("env_episode") is not None and hasattr(self.env, "episode"):
            self.env.episode = sd["env_episode"]
        np.random.set_state(sd["np_random"])

    def evaluate(self):
        """reference runner.py:115-121."""
        if self.args.evaluate_epoch == 0:
            return 0, 0
        n = getattr(self.env, "n_envs", self.args.evaluate_epoch)
        _, episodes_reward, win_tags, _ = self.rolloutWorker.generate_episodes(n_episodes=n, evaluate=True)
        return sum(win_tags) / len(win_tags), sum(episodes_reward) / len(episodes_reward)

    def plt(self, num):
        """reference runner.py:123-141: the curves are saved as .npy (the PNG needs matplotlib, optional)."""
        np.save(self.save_path + '/win_rates_{}'.format(num), self.eval_win_rates)
        np.save(self.save_path + '/episode_rewards_{}'.format(num), self.eval_episode_rewards)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.figure()
            plt.subplot(2, 1, 1); plt.plot(range(len(self.eval_win_rates)), self.eval_win_rates)
            plt.ylabel('win_rates')
            plt.subplot(2, 1, 2); plt.plot(range(len(self.eval_episode_rewards)), self.eval_episode_rewards)
            plt.xlabel('step*{}'.format(self.args.evaluate_cycle)); plt.ylabel('episode_rewards')
            plt.savefig(self.save_path + '/plt_{}.png'.format(num), format='png')
            plt.close()
        except Exception:
            pass
