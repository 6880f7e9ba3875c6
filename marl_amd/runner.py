"""Runner (mirror of reference runner.py:14-141): wires MAC + RolloutWorker + ReplayBuffer + learner
and reproduces the train / evaluate cadence, the logging tags and the save cycle.  With a batched env
every iteration collects ``env.n_envs`` episodes in lock-step and trains on ``batch_size`` sampled
episodes; with a serial env it behaves exactly like the reference loop."""
from __future__ import annotations

import os

import numpy as np

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

    def run(self, num):
        """reference runner.py:61-113."""
        a = self.args
        time_steps, train_steps, evaluate_steps = 0, 0, -1
        n_ep = getattr(self.env, "n_envs", a.n_episodes)
        loss = float("nan")
        while time_steps < a.n_steps:
            if time_steps // a.evaluate_cycle > evaluate_steps:
                win_rate, episode_reward = self.evaluate()
                self.eval_win_rates.append(win_rate)
                self.eval_episode_rewards.append(episode_reward)
                self.plt(num)
                self.logger.log_stat("test_win_rate", win_rate, time_steps)
                self.logger.log_stat("test_episode_reward", episode_reward, time_steps)
                evaluate_steps += 1
            episodes, rewards, win_tags, steps = self.rolloutWorker.generate_episodes(n_episodes=n_ep, random_select=False)
            time_steps += steps
            self.logger.log_stat("episode_length", steps, time_steps)
            self.logger.log_stat("train_win_rate", sum(win_tags) / n_ep, time_steps)
            self.logger.log_stat("train_episode_reward", sum(rewards) / n_ep, time_steps)
            self.buffer.store_episode(episodes)
            for _ in range(a.train_steps):
                mini_batch = self.buffer.sample(min(self.buffer.current_size, a.batch_size))
                loss = self.learner.train(mini_batch, train_steps)
                train_steps += 1
            self.logger.log_stat("total_loss", loss, time_steps)
            if train_steps > 0 and train_steps % a.save_cycle == 0:
                self.learner.save_models(train_steps)
        win_rate, episode_reward = self.evaluate()
        self.eval_win_rates.append(win_rate)
        self.eval_episode_rewards.append(episode_reward)
        self.plt(num)
        return loss

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
