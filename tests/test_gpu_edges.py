"""Edge and mid-size learner parity on the MI355X vs the CPU oracle (itself pinned by the golden vectors):
sizes the golden cases do not reach - several row tiles with a partial last one (the fused QPLEX head kernels
walk 16 / 64-row tiles), a single episode of a single step, every episode cut short, zero-row kernel calls.
Tolerance: losses relative 1e-4, gradients 1e-4 of their scale (north_star)."""
import numpy as np
import pytest
import torch

from oracle import seeded, learners

from test_gpu_learners import build_product, named_product_params
import parity

pytestmark = pytest.mark.gpu

#        name            shape   alg           B   T   lengths (None = seeded ragged, -1 = never terminates)
EDGE = [("qplex_multi_tile", "2s3z", "qplex", 37, 9, None, {}),
        ("qmix_multi_tile", "2s3z", "qmix", 50, 7, None, {}),
        ("qtran_multi_tile", "3s5z", "qtran_base", 11, 6, None, {}),
        ("qplex_one_step", "2s3z", "qplex", 1, 1, [1], {}),
        ("qmix_all_short", "2s3z", "qmix", 5, 8, [2, 1, 3, 2, 1], {}),
        ("vdn_unterminated", "2s3z", "vdn", 3, 4, [-1, -1, -1], {})]


@pytest.mark.parametrize("case", EDGE, ids=[c[0] for c in EDGE])
def test_train_vs_oracle(case):
    from golden_cases import build_oracle_state
    name, shape, alg, B, T, lengths, over = case
    if lengths is None:
        rng = np.random.default_rng(B + T)
        lengths = [int(x) for x in rng.integers(1, T + 1, size=B)]
        lengths[0] = T                       # at least one full-length episode
        lengths[-1] = -1                     # and one that never terminates (quirk Q2)
    case = (name, shape, alg, B, T, lengths, over)
    args, mac, learner = build_product(case)
    _, ost = build_oracle_state(case)
    for i, ts in enumerate((0, 1)):
        batch = seeded.make_batch(args, B, seed=300 + i, lengths=lengths)
        loss = learner.train(learners.clone_batch(batch), ts)
        oloss, ograds, ointer = learners.train(ost, learners.clone_batch(batch), ts)
        assert learner.max_episode_len == ointer["T"]
        parity.close("edge:" + name, "loss step %d" % i, loss, oloss, tol=1e-4 * (1 + 9 * i))
        if i == 0:
            den = float(learner.last_stats[-1 if alg.startswith("qtran") else 1].item())
            for n, p in named_product_params(learner):
                og = ograds.get(n)
                g = p.grad.detach().cpu().numpy() / den
                if og is None:
                    assert np.all(g == 0), n
                    continue
                parity.close("edge:" + name, "grad " + n, g, og.detach().numpy(), tol=1e-4)


def test_zero_rows_are_noops():
    """Empty inputs return success and touch nothing (the reference's loops simply do not run)."""
    from marl_amd import ops
    dev = torch.device("cuda:0")
    x = torch.zeros(4, 16, device=dev)
    W, b = torch.randn(8, 16, device=dev), torch.randn(8, device=dev)
    Y = torch.full((4, 8), 3.0, device=dev)
    ops.linear(ops.src(x), W, b, Y, 0, 8, 16)
    dW, db = torch.full((8, 16), 2.0, device=dev), torch.full((8,), 2.0, device=dev)
    ops.linear_wgrad(Y, ops.src(x), dW, db, 0, 8, 16)
    out = torch.full((4,), 5.0, device=dev)
    ops.agent_sum(Y, out, 0, 8, 1)
    ops.vec_add(out, out, out, 0)
    torch.cuda.synchronize()
    assert float(Y.min()) == 3.0 and float(dW.min()) == 2.0 and float(db.min()) == 2.0 and float(out.min()) == 5.0


def test_first_terminated_len_kernel_matches_reference_rule():
    """marl_first_terminated_len vs the reference's get_max_episode_len rule (q_learner.py:49-66, quirk Q2)."""
    from marl_amd import ops
    from marl_amd.hostutil import DeviceBatch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for E, T in ((1, 1), (7, 5), (300, 120), (4096, 150), (5, 64), (3, 65)):
        term = np.zeros((E, T, 1), np.float32)
        lens = rng.integers(0, T + 1, size=E)              # 0 = never terminates
        for e, L in enumerate(lens):
            if L > 0:
                term[e, L - 1:, 0] = 1.0
        ref = 0
        for e in range(E):                                 # the reference's loop
            for t in range(T):
                if term[e, t, 0] == 1:
                    ref = max(ref, t + 1)
                    break
        got = int(ops.first_terminated_len(torch.as_tensor(term).to(dev), T).item())
        assert got == ref, (E, T, got, ref)
        assert DeviceBatch.first_terminated_len(torch.as_tensor(term).to(dev), T) == (ref if ref > 0 else T)
        assert DeviceBatch.first_terminated_len(torch.as_tensor(term), T) == (ref if ref > 0 else T)      # host path
    none = torch.zeros(4, 9, device=dev)
    assert DeviceBatch.first_terminated_len(none, 9) == 9


@pytest.mark.parametrize("alg,gemm_mode", [("qmix", "f32"), ("qplex", "f32"), ("qtran_base", "f32"), ("qmix", "bf16x6"), ("qplex", "bf16x6")])
def test_hip_graph_replay_equals_eager(alg, gemm_mode):
    """hipGraph replay of the learner's forward/backward (args.hip_graph: the default for small batches): same ring, same sampled
    episodes -> bitwise the same losses and parameters as eager launches, across the capture (update 3), the replays and the
    speculative replays (from update 5 on the graph is launched before max_episode_len is read back).  QPLEX / QTRAN read the
    current-step availability of the SAMPLED episodes through the batch's index tensor (a different sample per update: the
    replayed gather must follow it)."""
    import bench
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd.common.replaybuffer import ReplayBuffer
    out = {}
    for mode in (False, True):
        args = bench.make_args(alg, "2s3z", 12)
        E = 96
        args.buffer_size, args.batch_size, args.hip_graph, args.gemm_mode = 2 * E, E, mode, gemm_mode
        torch.manual_seed(0)
        np.random.seed(7)
        mac = SharedMAC(args)
        from marl_amd.algorithm.qtran_learner import QTRANLearner
        learner = QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args)
        env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, 12, seed=3, fixed_length=True)
        w = RolloutWorker(env, mac, args)
        buf = ReplayBuffer(args)
        w.record_sink = buf
        losses = []
        for i in range(8):
            ep = w.generate_episodes(E)[0]
            buf.store_episode(ep)
            losses.append(learner.train(buf.sample(E), i))
        out[mode] = (losses, learner._flat.flat.detach().cpu().numpy().copy())
        if mode:
            g = learner.graphs
            assert not g.disabled, getattr(g, "error", "")
            assert any(e["graph"] is not None for e in g.entries.values()), "no graph was captured"
            assert g.replays >= 5 and all(e["streak"] >= 2 for e in g.entries.values())
        else:
            assert learner.graphs is None
    assert out[False][0] == out[True][0]
    np.testing.assert_array_equal(out[False][1], out[True][1])


def test_speculative_graph_replay_redoes_a_short_batch():
    """the same rule on the hipGraph path: once replays run before max_episode_len is read back, a ring whose sampled episodes
    ALL ended early gets the (already replayed) update redone eagerly at its own length - same losses, lengths and parameters
    as a learner without graphs"""
    import bench
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd.common.replaybuffer import ReplayBuffer
    out = {}
    for mode in (False, True):
        args = bench.make_args("qmix", "2s3z", 12)
        E = 48
        args.buffer_size, args.batch_size, args.hip_graph = E, E, mode
        torch.manual_seed(0)
        np.random.seed(3)
        mac = SharedMAC(args)
        learner = QLearner(mac, args)
        env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, 12, seed=5, fixed_length=True)
        w = RolloutWorker(env, mac, args)
        buf = ReplayBuffer(args)
        w.record_sink = buf
        losses, lens = [], []
        for i in range(9):
            ep = w.generate_episodes(E)[0]
            buf.store_episode(ep)
            if i == 7:                          # every stored episode ends after 5 steps
                rec = buf.record
                rec.term[:, 4:] = 1.0
                rec.padded[:, 5:] = 1.0
                rec.length.fill_(5)
            losses.append(learner.train(buf.sample(E), i))
            lens.append(learner.max_episode_len)
        out[mode] = (losses, lens, learner._flat.flat.detach().cpu().numpy().copy())
        if mode:
            assert learner.graphs.replays >= 4
    assert out[True][1] == out[False][1] == [12] * 7 + [5, 12]
    assert out[True][0] == out[False][0]
    np.testing.assert_array_equal(out[True][2], out[False][2])


def test_speculative_full_length_launch_redoes_a_short_batch():
    """QLearner launches forward / backward for the record's full length before max_episode_len is read back once the
    last updates all ran at full length; a batch whose episodes ALL ended early must then be redone at its own length -
    same losses and parameters as a learner that always reads the length first."""
    import bench
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    out = {}
    for spec in (False, True):
        args = bench.make_args("qmix", "2s3z", 12)
        E = 40
        torch.manual_seed(0)
        mac = SharedMAC(args)
        learner = QLearner(mac, args)
        env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, 12, seed=5, fixed_length=True)
        w = RolloutWorker(env, mac, args)
        losses, lens = [], []
        for i in range(6):
            ep = w.generate_episodes(E)[0]
            if i == 4:                          # every episode of this batch ends after 5 steps
                rec = ep.record
                rec.term[:, 4:] = 1.0
                rec.padded[:, 5:] = 1.0
                rec.length.fill_(5)
            if not spec:
                learner._full_len_streak = 0     # never speculate
            losses.append(learner.train(ep, i))
            lens.append(learner.max_episode_len)
        out[spec] = (losses, lens, learner._flat.flat.detach().cpu().numpy().copy())
    assert out[True][1] == out[False][1] == [12, 12, 12, 12, 5, 12]
    assert out[True][0] == out[False][0]
    np.testing.assert_array_equal(out[True][2], out[False][2])


@pytest.mark.parametrize("shape", ["2s3z", "MMM2"])
def test_qmix_loss_folded_into_the_mixer_backward(shape):
    """QMIX on the fused mixer kernels (2s3z: registers-resident hypernet; MMM2: wide-state kernel): eval-mixer forward + TD
    loss + mixer backward as ONE launch (the backward recomputes q_tot anyway) against the three-launch path
    (args.no_loss_fold) - same loss, sum(mask), gradients and q_tot up to the summation order of the kernels' partial sums."""
    case = ("qmix_fold", shape, "qmix", 37, 9, None, {})
    name, shape, alg, B, T, lengths, over = case
    rng = np.random.default_rng(5)
    lengths = [int(x) for x in rng.integers(1, T + 1, size=B)]
    lengths[0], lengths[-1] = T, -1
    case = (name, shape, alg, B, T, lengths, over)
    out = {}
    for fold in (True, False):
        args, mac, learner = build_product(case)
        args.no_loss_fold = not fold
        batch = seeded.make_batch(args, B, seed=11, lengths=lengths)
        loss = learner.train(learners.clone_batch(batch), 0)
        out[fold] = (loss, learner.last_stats[:2].cpu().numpy().copy(), learner._flat.gradx.detach().cpu().numpy().copy(),
                     learner._dbg["q_tot"].detach().cpu().numpy().copy())
    assert out[True][1][1] == out[False][1][1]                         # sum(mask): exact
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-6)
    np.testing.assert_allclose(out[True][3], out[False][3], rtol=0, atol=2e-6 * np.abs(out[False][3]).max())
    scale = np.abs(out[False][2]).max()
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=0, atol=2e-6 * scale)
