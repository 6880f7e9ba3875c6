"""smac-free launcher (counterpart of reference main.py:7-44):

    python -m marl_amd.main --alg qmix --map 2s3z --n_envs 1024 --n_steps 500000
    python -m marl_amd.main --env matrix --alg qplex --n_envs 32 --n_steps 20000

``--env synthetic`` (default) uses the synthetic SMAC-shaped device env with the dims of ``--map``;
``--env matrix`` the batched two-agent matrix game.  StarCraft II itself is not vendored by the reference."""
from __future__ import annotations

import sys

from .common.arguments import get_common_args, get_mixer_args
from .env.synthetic_smac import SyntheticSMACEnv
from .env.single_state_matrix_game import BatchedMatrixGame
from .runner import Runner
from .utils.logging import Logger

MAPS = {"2s3z": (5, 80, 120, 11, 120), "3s5z": (8, 128, 216, 14, 150), "MMM2": (10, 176, 322, 18, 120)}


def build(argv=None):
    args = get_common_args(argv)
    get_mixer_args(args)
    if args.env == 'smac':
        args.env = 'synthetic'
    if args.env == 'synthetic':
        n, o, s, a, t = MAPS[args.map]
        env = SyntheticSMACEnv(args.n_envs, n, o, s, a, t, seed=args.seed)
    elif args.env == 'matrix':
        env = BatchedMatrixGame([[8, -12, -12], [-12, 0, 0], [-12, 0, 0]], args.n_envs, seed=args.seed)
        args.map = 'MatrixGame'
    else:
        raise ValueError("env not found")
    info = env.get_env_info()
    args.n_actions, args.n_agents = info["n_actions"], info["n_agents"]
    args.state_shape, args.obs_shape, args.episode_limit = info["state_shape"], info["obs_shape"], info["episode_limit"]
    args.batch_size = max(args.batch_size, args.n_envs)
    args.buffer_size = max(2 * args.n_envs, min(args.buffer_size, 8 * args.n_envs))
    return args, env


def main(argv=None):
    args, env = build(argv)
    runner = Runner(env, Logger(), args)
    if not args.evaluate:
        loss = runner.run(0)
        print("final loss", loss)
    else:
        win_rate, _ = runner.evaluate()
        print('The win rate of {} is  {}'.format(args.alg, win_rate))
    env.close()


if __name__ == '__main__':
    main(sys.argv[1:])
