"""Argument helpers (mirror of reference common/arguments.py:9-45, 86-147) restricted to the fields
the hot path reads.  Booleans parse properly here (the reference's ``type=bool`` treats any
non-empty string as True)."""
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
    args.adv_hypernet_embed = 64
    args.num_kernel = 10
    args.adv_hypernet_layers = 3
    args.weighted_head = True
    args.hypernet_embed = 64
    args.is_minus_one = True
    args.mixing_embed_dim = 32
    args.double_q = True
    return args
