"""Host-side logic that needs no GPU: replay ring arithmetic vs the reference fixture, the device
form of get_max_episode_len, argument parsing, drop-in module paths."""
import os
import sys

import numpy as np
import torch

from oracle import seeded, learners


def test_replay_host_mode_matches_reference(golden_dir):
    from marl_amd.common.replaybuffer import ReplayBuffer
    fix = np.load(os.path.join(golden_dir, "replay.npz"))
    args = seeded.make_args("2s3z", "qmix", episode_limit=3, buffer_size=7)
    buf = ReplayBuffer(args)
    log = []
    for i, n in enumerate([1, 3, 2, 3, 1, 7, 2]):
        before = buf.current_idx
        buf.store_episode(seeded.make_batch(args, n, seed=300 + i))
        log.append([before, buf.current_idx, buf.current_size])
    np.testing.assert_array_equal(np.array(log), fix["state_log"])
    np.testing.assert_array_equal(buf.buffers["r"], fix["final_r"])
    np.random.seed(21)
    s = buf.sample(5)
    np.testing.assert_array_equal(s["r"], fix["sample_r"])
    np.testing.assert_array_equal(s["u"], fix["sample_u"])
    assert set(s) == {"o", "u", "s", "r", "o_next", "s_next", "avail_u", "avail_u_next", "u_onehot", "padded", "terminated"}


def test_first_terminated_len_matches_reference_rule():
    from marl_amd.hostutil import DeviceBatch, onehot_to_index
    rng = np.random.default_rng(0)
    for _ in range(50):
        B, T = int(rng.integers(1, 6)), int(rng.integers(1, 9))
        term = (rng.random((B, T, 1)) < 0.25).astype(np.float64)
        assert DeviceBatch.first_terminated_len(torch.tensor(term), T) == learners.max_episode_len(term, T)
    assert DeviceBatch.first_terminated_len(torch.zeros(3, 5, 1), 5) == 5      # quirk Q2: none terminated
    oh = torch.tensor([[0., 1, 0], [0, 0, 0], [0, 0, 1]])
    assert onehot_to_index(oh).tolist() == [1, -1, 2]


def test_arguments_and_dropin_paths():
    from marl_amd.common.arguments import get_common_args, get_mixer_args
    a = get_mixer_args(get_common_args(["--cuda", "False", "--alg", "qplex"]))
    assert a.cuda is False and a.alg == "qplex"          # the reference would parse "False" as True
    assert a.rnn_hidden_dim == 64 and a.qmix_hidden_dim == 32 and a.double_q and a.target_update_cycle == 200
    assert abs(a.anneal_epsilon - 0.95 / 50000) < 1e-15
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "marl_amd", "dropin"))
    try:
        for m in ("rollout", "controller", "algorithm", "common", "network", "env"):
            sys.modules.pop(m, None)
        from rollout import RolloutWorker                                        # noqa: F401
        from controller.share_params import SharedMAC, SeparatedMAC, SharedMACWithState, RTWMAC   # noqa: F401
        from common.replaybuffer import ReplayBuffer                             # noqa: F401
        from algorithm.q_learner import QLearner                                 # noqa: F401
        from algorithm.qtran_learner import QTRANLearner                         # noqa: F401
        from algorithm.RTW_q_learner import RTWQLearner                          # noqa: F401
        from algorithm.q_learner_state import QLearnerWithState                  # noqa: F401
        from network.mixer import VDNMixer, QMixMixer, DMAQer, QtranQBase, QtranQAlt, QtranV   # noqa: F401
        from network.q_network import RNNQNet                                    # noqa: F401
        from env.single_state_matrix_game import TwoAgentsMatrixGame             # noqa: F401
    finally:
        sys.path.pop(0)
        for m in list(sys.modules):
            if m.split(".")[0] in ("rollout", "controller", "algorithm", "common", "network", "env"):
                sys.modules.pop(m, None)


def test_matrix_game_get_episodes_matches_reference(golden_dir):
    from marl_amd.env.single_state_matrix_game import TwoAgentsMatrixGame
    fix = np.load(os.path.join(golden_dir, "rollout.npz"))
    env = TwoAgentsMatrixGame([[8, -12, -12], [-12, 0, 0], [-12, 0, 0]])
    for k, v in env.get_episodes().items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), fix["matrix_get_episodes/" + k])
    r, term, info = env.step([0, 0])
    assert r == 8 and term is True


def test_state_dicts_match_reference_checkpoints(golden_dir):
    """Checkpoint interop: the key names and tensor shapes of the product modules equal
    those of the .pkl files the reference ships under model/{vdn,qplex,qtran_base}/2s3z (table
    extracted by torch.load in the build container; model/qmix/2s3z/*rnn* are RTW-agent files and
    are skipped).  Modules are constructed on CPU - no compute is called."""
    import json
    from marl_amd.network.q_network import RNNQNet
    from marl_amd.network.mixer import QMixMixer, DMAQer, QtranQBase, QtranV, VDNMixer
    table = json.load(open(os.path.join(golden_dir, "reference_checkpoint_shapes.json")))
    def shapes(m):
        return {k: list(v.shape) for k, v in m.state_dict().items()}
    checked = 0
    for path, ref in table.items():
        alg = path.split("/")[1]
        args = seeded.make_args("2s3z", alg)
        kind = os.path.basename(path)
        if "rnn_net" in kind:
            if alg == "qmix":
                continue                                    # RTW agent checkpoints (16 extra keys)
            got = shapes(RNNQNet(96, args))
        elif "v_net" in kind:
            got = shapes(QtranV(args))
        else:
            got = shapes({"vdn": VDNMixer, "qmix": QMixMixer, "qplex": DMAQer, "qtran_base": QtranQBase}[alg](args))
        assert got == ref, path
        checked += 1
    assert checked >= 20
