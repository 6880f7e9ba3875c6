"""Argument helpers (mirror of reference common/arguments.py:9-45, 86-147): every function the reference's callers
import (matrix_game_test.py:9, main.py:3) exists with the same name and effect.  Booleans parse properly here (the
reference's ``type=bool`` treats any non-empty string as True), and ``--RTW`` / ``--load_model``
default to False: the RTW research variant is outside the hot path and no checkpoint ships with this build."""
import argparse


def _bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() not in ("", "0", "false", "no", "none")


def get_common_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--RTW', type=_bool, default=False)
    p.add_argument('--env', type=str, default='smac')
    p.add_argument('--difficulty', type=str, default='7')
    p.add_argument('--game_version', type=str, default='latest')
    p.add_argument('--map', type=str, default='2s3z')
    p.add_argument('--seed', type=int, default=123)
    p.add_argument('--step_mul', type=int, default=8)
    p.add_argument('--replay_dir', type=str, default='')
    p.add_argument('--alg', type=str, default='qmix')
    p.add_argument('--n_steps', type=int, default=800000)
    p.add_argument('--n_episodes', type=int, default=1)
    p.add_argument('--last_action', type=_bool, default=True)
    p.add_argument('--reuse_network', type=_bool, default=True)
    p.add_argument('--gamma', type=float, default=0.99)
    p.add_argument('--optimizer', type=str, default="RMS")
    p.add_argument('--evaluate_cycle', type=int, default=5000)
    p.add_argument('--evaluate_epoch', type=int, default=0)
    p.add_argument('--model_dir', type=str, default='./model')
    p.add_argument('--result_dir', type=str, default='./result')
    p.add_argument('--load_model', type=_bool, default=False)
    p.add_argument('--evaluate', type=_bool, default=False)
    p.add_argument('--cuda', type=_bool, default=True)
    p.add_argument('--n_envs', type=int, default=1, help='parallel environments for the batched rollout')
    return p.parse_args(argv)


def get_mixer_args(args):
    args.rnn_hidden_dim = 64
    args.qmix_hidden_dim = 32
    args.two_hyper_layers = False
    if not hasattr(args, "mixer_dtype"):
        args.mixer_dtype = "fp32"     # build extension (BASELINE config 5): "bf16" = mixer GEMMs on the bf16 matrix cores
    args.hyper_hidden_dim = 64
    args.qtran_hidden_dim = 64
    args.lr = 5e-4
    args.epsilon = 1
    args.min_epsilon = 0.05
    anneal_steps = 50000
    args.anneal_epsilon = (args.epsilon - args.min_epsilon) / anneal_steps
    args.epsilon_anneal_scale = 'step'
    args.train_steps = 1
    args.batch_size = 32
    args.buffer_size = int(5e3)
    args.save_cycle = 5000
    args.target_update_cycle = 200
    args.lambda_opt = 1
    args.lambda_nopt = 1
    args.grad_norm_clip = 10
    args.noise_dim, args.lambda_mi, args.lambda_ql, args.entropy_coefficient = 16, 0.001, 1, 0.001   # MAVEN (unused here)
    args.adv_hypernet_embed = 64
    args.num_kernel = 10
    args.adv_hypernet_layers = 3
    args.weighted_head = True
    args.hypernet_embed = 64
    args.is_minus_one = True
    args.mixing_embed_dim = 32
    args.double_q = True
    return args


# ---- hyper-parameter tables of the algorithms outside the hot path (reference common/arguments.py:48-83,151-214).
# Nothing in marl_amd reads these fields; the functions exist because the reference's entry scripts import them
# (matrix_game_test.py:9, main.py:3) and main.py:12 calls get_RTW_args on every run.
_ACTOR_CRITIC = dict(rnn_hidden_dim=64, critic_dim=128, lr_actor=1e-4, lr_critic=1e-3, epsilon=0.5,
                     anneal_epsilon=0.00064, min_epsilon=0.02, epsilon_anneal_scale='episode', save_cycle=5000,
                     grad_norm_clip=10)


def _assign(args, table):
    for k, v in table.items():
        setattr(args, k, v)
    return args


def get_RTW_args(args):
    """reference :48-53 (returns None there as well)"""
    _assign(args, dict(world_loss_weight=1, teammate_loss_weight=1, hidden_dim=64, attn_dim=64, not_self_model=True))


def get_coma_args(args):
    """reference :56-83"""
    return _assign(args, dict(_ACTOR_CRITIC, td_lambda=0.8, target_update_cycle=200))


def get_centralv_args(args):
    """reference :151-177"""
    return _assign(args, dict(_ACTOR_CRITIC, td_lambda=0.8, target_update_cycle=200))


def get_reinforce_args(args):
    """reference :181-200"""
    return _assign(args, dict(_ACTOR_CRITIC))


def get_commnet_args(args):
    """reference :204-209"""
    args.k = 2 if args.map == '3m' else 3
    return args


def get_g2anet_args(args):
    """reference :212-215"""
    return _assign(args, dict(attention_dim=32, hard=True))
