"""Matrix-game harness (mirror of reference matrix_game_test.py:35-113): trains on the fixed
9-episode batch of all joint actions and prints the learned q_tot table / greedy joint action."""
from __future__ import annotations

import types

import numpy as np

from .common.arguments import get_mixer_args
from .controller.share_params import SharedMAC
from .algorithm.q_learner import QLearner
from .algorithm.qtran_learner import QTRANLearner
from .env.single_state_matrix_game import TwoAgentsMatrixGame

PAYOFF1 = [[8, -12, -12], [-12, 0, 0], [-12, 0, 0]]


def make_args(alg, lr=0.001):
    a = types.SimpleNamespace(alg=alg, map="MatrixGame", last_action=True, reuse_network=True, gamma=0.99,
                              optimizer="RMS", cuda=True, RTW=False, load_model=False, model_dir="./model",
                              result_dir="./result", replay_dir="", n_episodes=1, evaluate_epoch=0, seed=123)
    get_mixer_args(a)
    a.lr = lr
    return a


def run(alg="qtran_base", tot_epoch=20000, payoff=PAYOFF1, verbose=True):
    args = make_args(alg)
    env = TwoAgentsMatrixGame(payoff_table=payoff)
    info = env.get_env_info()
    args.n_actions, args.n_agents = info["n_actions"], info["n_agents"]
    args.state_shape, args.obs_shape, args.episode_limit = info["state_shape"], info["obs_shape"], info["episode_limit"]
    mac = SharedMAC(args)
    learner = QTRANLearner(mac, args) if alg == 'qtran_base' else QLearner(mac, args)
    loss = None
    for it in range(tot_epoch):
        loss = learner.train(env.get_episodes(), it)
        if verbose and (it + 1) % 1000 == 0:
            print('Iteration: {a}   MSE loss: {b}'.format(a=it + 1, b=loss))
    q_tot_table, q_table_i, q_table_j = learner.get_q_and_q_tot_table()
    r, c = divmod(int(q_tot_table.argmax()), q_tot_table.shape[1])
    if verbose:
        print("q_tot_table\n", q_tot_table, "\ngreedy joint-action is: ", [r, c])
        print("q_i", q_table_i, "q_j", q_table_j)
    greedy_individual = [int(np.argmax(q_table_i)), int(np.argmax(q_table_j))]
    return q_tot_table, [r, c], greedy_individual, loss


if __name__ == '__main__':
    import sys
    run(sys.argv[1] if len(sys.argv) > 1 else "qtran_base", int(sys.argv[2]) if len(sys.argv) > 2 else 20000)
