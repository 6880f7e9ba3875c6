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


DROPIN_TOP = ("rollout", "runner", "controller", "algorithm", "common", "network", "env", "utils", "smac")


def _forget_dropin_modules():
    for m in list(sys.modules):
        if m.split(".")[0] in DROPIN_TOP:
            sys.modules.pop(m, None)


def test_arguments_match_reference_tables():
    from marl_amd.common.arguments import (get_common_args, get_mixer_args, get_coma_args, get_centralv_args,
                                           get_reinforce_args, get_commnet_args, get_g2anet_args, get_RTW_args)
    a = get_mixer_args(get_common_args(["--cuda", "False", "--alg", "qplex"]))
    assert a.cuda is False and a.alg == "qplex"          # the reference would parse "False" as True
    assert a.rnn_hidden_dim == 64 and a.qmix_hidden_dim == 32 and a.double_q and a.target_update_cycle == 200
    assert abs(a.anneal_epsilon - 0.95 / 50000) < 1e-15
    assert a.RTW is False and a.load_model is False
    assert get_RTW_args(a) is None and a.attn_dim == 64 and a.not_self_model is True     # reference :48-53
    b = get_coma_args(get_common_args([]))
    assert (b.critic_dim, b.lr_actor, b.lr_critic, b.td_lambda, b.epsilon_anneal_scale) == (128, 1e-4, 1e-3, 0.8, 'episode')
    c = get_centralv_args(get_common_args([]))
    assert c.td_lambda == 0.8 and c.target_update_cycle == 200 and c.anneal_epsilon == 0.00064
    d = get_reinforce_args(get_common_args([]))
    assert not hasattr(d, "td_lambda") and d.min_epsilon == 0.02 and d.grad_norm_clip == 10
    assert get_commnet_args(get_common_args(["--map", "3m"])).k == 2 and get_commnet_args(get_common_args([])).k == 3
    g = get_g2anet_args(get_common_args([]))
    assert g.attention_dim == 32 and g.hard is True


def test_reference_import_lines_resolve_to_dropin(golden_dir):
    """The LITERAL import statements of the reference's callers (runner.py:3-11, matrix_game_test.py:3-10,
    main.py:1-5; committed as data in tests/golden/reference_import_lines.json) executed with marl_amd/dropin first on
    sys.path, exactly as `python -m marl_amd.dropin <script>` arranges it: every hot-path name must come from marl_amd.
    `smac` (StarCraft II, not vendored by the reference) resolves to the synthetic-env shim when the real package is
    not installed."""
    import json
    from marl_amd.dropin.__main__ import install
    table = json.load(open(os.path.join(golden_dir, "reference_import_lines.json")))
    saved_path = list(sys.path)
    pkg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marl_amd") + os.sep
    _forget_dropin_modules()
    try:
        install()
        for script, ent in table.items():
            ns = {}
            for line in ent["imports"]:
                exec(line, ns)          # raises ImportError if a name is missing
            for name, obj in ns.items():
                mod = getattr(obj, "__module__", None) or getattr(obj, "__name__", "")
                if name in ("plt", "np", "os", "__builtins__"):
                    continue
                f = os.path.abspath(getattr(sys.modules.get(str(mod)), "__file__", "") or "")
                assert f.startswith(pkg_dir), (script, name, mod, f)      # defined inside marl_amd/ (dropin stubs included)
        # what main.py:10-12 and matrix_game_test.py:36-37 call right after the imports
        ns = {}
        exec("from common.arguments import get_common_args, get_mixer_args, get_RTW_args", ns)
        argv, sys.argv = sys.argv, ["x"]
        try:
            args = ns["get_common_args"]()
        finally:
            sys.argv = argv
        ns["get_mixer_args"](args)
        ns["get_RTW_args"](args)
        assert args.RTW is False and args.world_loss_weight == 1
    finally:
        sys.path[:] = saved_path
        _forget_dropin_modules()


def test_reference_matrix_game_script_runs_on_dropin():
    """With the reference checkout present (build container only): its own matrix_game_test.py, unchanged, through the
    launcher.  Without a GPU the run must get as far as the learner's constructor (every import, get_common_args,
    the env, SharedMAC and RolloutWorker resolved to marl_amd) and stop at "no CPU fallback"; with a GPU the full
    20 000-iteration script is the job of tests/test_gpu_runner.py's shorter harness."""
    import subprocess
    script = "/root/reference/matrix_game_test.py"
    if not os.path.exists(script):
        import pytest
        pytest.skip("reference checkout not present")
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by the GPU harness")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:      # the script writes ./result/...
        p = subprocess.run([sys.executable, "-m", "marl_amd.dropin", script], cwd=tmp, capture_output=True, text=True,
                           env=dict(os.environ, PYTHONPATH=root, MPLBACKEND="Agg"), timeout=300)
    assert p.returncode != 0
    assert "Init RolloutWorker" in p.stdout                              # marl_amd.rollout.RolloutWorker.__init__
    assert "QTRANLearner runs only on the MI355X HIP kernels (no CPU fallback)" in p.stderr, p.stderr[-2000:]
    assert "marl_amd/algorithm/qtran_learner.py" in p.stderr and "/root/reference/rollout.py" not in p.stderr


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


def test_launcher_vectorises_an_importable_smac(tmp_path, monkeypatch):
    """`python -m marl_amd.dropin` with a real `smac` on the path: MARL_N_ENVS > 1 swaps smac.env.StarCraft2Env for a
    factory of HostVectorEnv over that many real environments (not called here - it needs the GPU); <= 1 leaves it alone."""
    pkg = tmp_path / "smac"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "env.py").write_text("class StarCraft2Env:\n    pass\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    _forget_dropin_modules()
    from marl_amd.dropin.__main__ import vectorise_real_smac
    import smac.env as se
    real = se.StarCraft2Env
    assert vectorise_real_smac(1) is False and se.StarCraft2Env is real
    assert vectorise_real_smac(8) is True
    assert se.StarCraft2Env is not real and se.StarCraft2Env._marl_real is real
    assert vectorise_real_smac(8) is True and se.StarCraft2Env._marl_real is real      # idempotent
    _forget_dropin_modules()
