"""Rollout / replay parity on the MI355X: batched device rollout vs the CPU oracle (bit-exact
integer/record fields, 1e-4 on floats), serial compat path vs the reference fixtures, and the
record-backed learner path vs the dict path."""
import numpy as np
import pytest
import torch

from oracle import seeded, rollout as orl, learners

pytestmark = pytest.mark.gpu


def _mac(args, scale=3.0):
    from marl_amd.controller.share_params import SharedMAC
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11, scale=scale)
    mac = SharedMAC(args)
    mac.agent.load_state_dict({k: torch.tensor(v) for k, v in agent.items()})
    mac.cuda()
    return mac, agent


@pytest.mark.parametrize("eps,evaluate", [(0.0, True), (0.5, False), (1.0, False)])
def test_batched_rollout_matches_oracle(eps, evaluate, gemm_mode):
    """(gemm_mode "bf16x6": the whole-rollout kernel is csrc/rollout_x6.hip - the agent step as split products; the per-step paths
    it is compared with run the fp32 kernels: the integer fields agree because no two available actions' Q values of these cases
    lie within rounding of each other)"""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd import ops
    T, E = 8, 37
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=eps, seed=77)
    args.anneal_epsilon = 0.01
    args.gemm_mode = gemm_mode
    assert ops.synth_rollout_x6_supported(5, 80, 11)
    mac, agent = _mac(args)
    env = SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=5, env0=2)
    w = RolloutWorker(env, mac, args)
    ep, rew, wins, steps = w.generate_episodes(E, evaluate=evaluate)
    # the whole-rollout kernel wrote the per-episode statistics itself (reward sums | won | length; compared with the
    # oracle's below) - and they agree with a reduction of the record it wrote
    assert getattr(env, "_stats_ring", None) and ep.record.kernel_stats is None
    np.testing.assert_allclose(rew, ep.record.r.sum(1).cpu().numpy(), atol=1e-5)
    assert steps == int(ep.record.length.sum().item()) and list(wins) == [bool(x) for x in ep.record.won.cpu().tolist()]
    # three device paths, one record: whole-rollout persistent kernel (default), one fused env kernel per
    # lock-step, and the separate select / step / observe kernels
    for mode in ("fused_step", "unfused"):
        w2 = RolloutWorker(SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=5, env0=2), mac, args)
        w2.rollout_mode = mode
        ep_u, _, _, steps_u = w2.generate_episodes(E, evaluate=evaluate)
        assert steps_u == steps, mode
        for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
            assert torch.equal(getattr(ep.record, f), getattr(ep_u.record, f)), (mode, f)
        np.testing.assert_allclose(w2.epsilon, w.epsilon, rtol=1e-12)
    sy = orl.SynthSMAC(5, 80, 120, 11, T, seed=5)
    oep, orew, owins, osteps, oeps = orl.batched_rollout(agent, args, sy, E, eps, evaluate=evaluate, rseed=77, env0=2)
    got = ep.numpy()
    for k in ("u", "padded", "terminated", "avail_u", "avail_u_next", "u_onehot"):
        np.testing.assert_array_equal(got[k], np.asarray(oep[k], dtype=got[k].dtype), err_msg=k)
    for k in ("o", "o_next", "s", "s_next", "r"):
        np.testing.assert_allclose(got[k], oep[k], atol=1e-6, err_msg=k)
    assert steps == osteps and list(wins) == [bool(x) for x in owins]
    np.testing.assert_allclose(rew, orew, atol=1e-5)
    np.testing.assert_allclose(w.epsilon, oeps if not evaluate else eps, rtol=1e-12)
    # second rollout continues the env's episode counter and the epsilon schedule
    ep2, _, _, steps2 = w.generate_episodes(E, evaluate=evaluate)
    oep2, _, _, osteps2, _ = orl.batched_rollout(agent, args, sy, E, w.epsilon if evaluate else oeps, evaluate=evaluate,
                                                 rseed=77, env0=2, episode=1)
    assert steps2 == osteps2
    np.testing.assert_array_equal(ep2.numpy()["u"], oep2["u"])


def test_serial_rollout_matches_reference_fixture(golden_dir):
    """drop-in serial path (reference RNG order) vs RolloutWorker of the reference itself."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.single_state_matrix_game import TwoAgentsMatrixGame
    fix = np.load(golden_dir + "/rollout.npz")
    args = seeded.make_args("matrix", "vdn")
    mac, _ = _mac(args, scale=1.0)
    env = TwoAgentsMatrixGame([[8, -12, -12], [-12, 0, 0], [-12, 0, 0]])
    for tag, eps in (("eps1", 1.0), ("eps03", 0.3)):
        args.epsilon = eps
        w = RolloutWorker(env, mac, args)
        np.random.seed(7)
        ep, rew, wins, steps = w.generate_episodes(32)
        for k, v in ep.items():
            np.testing.assert_allclose(np.asarray(v, dtype=np.float64), fix["matrix_%s/%s" % (tag, k)], atol=1e-6, err_msg=k)
        assert steps == int(fix["matrix_%s/steps" % tag])
        np.testing.assert_allclose(w.epsilon, float(fix["matrix_%s/eps_after" % tag]), rtol=1e-12)
    ge = env.get_episodes()
    for k, v in ge.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), fix["matrix_get_episodes/" + k])
    # SMAC-shaped serial env
    args = seeded.make_args("2s3z", "qmix", episode_limit=8)
    mac, _ = _mac(args)
    for tag, eps, evaluate in (("greedy", 0.0, True), ("eps05", 0.5, False)):
        sy = orl.SynthSMAC(5, 80, 120, 11, 8, seed=5)
        args.epsilon = eps
        w = RolloutWorker(orl.SerialSynthEnv(sy), mac, args)
        np.random.seed(9)
        ep, rew, wins, steps = w.generate_episodes(6, evaluate=evaluate)
        for k in ("u", "r", "padded", "terminated", "avail_u", "avail_u_next"):
            np.testing.assert_allclose(np.asarray(ep[k], dtype=np.float64), fix["smac_%s/%s" % (tag, k)], atol=1e-6, err_msg=k)
        assert steps == int(fix["smac_%s/steps" % tag])


def test_batched_matrix_game_config1():
    """BASELINE config 1: VDN, matrix game, 32 parallel envs - rollout + train on the device."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.single_state_matrix_game import BatchedMatrixGame
    from marl_amd.algorithm.q_learner import QLearner
    payoff = [[8, -12, -12], [-12, 0, 0], [-12, 0, 0]]
    args = seeded.make_args("matrix", "vdn", epsilon=0.5, seed=3)
    mac, agent = _mac(args, scale=1.0)
    env = BatchedMatrixGame(payoff, 32)
    w = RolloutWorker(env, mac, args)
    learner = QLearner(mac, args)
    ep, rew, wins, steps = w.generate_episodes(32)
    d = ep.numpy()
    assert steps == 32 and d["o"].shape == (32, 1, 2, 1) and (d["terminated"] == 1).all() and (d["padded"] == 0).all()
    np.testing.assert_allclose(d["r"][:, 0, 0], np.array(payoff)[d["u"][:, 0, 0, 0], d["u"][:, 0, 1, 0]])
    st = learners.LearnerState(args, agent, {})
    oloss, _, _ = learners.train(st, d, 0)
    loss = learner.train(ep, 0)
    np.testing.assert_allclose(loss, oloss, rtol=1e-4)


@pytest.mark.parametrize("alg,shape", [("qmix", "2s3z"), ("qplex", "2s3z"), ("qtran_base", "3s5z")])
def test_record_path_equals_dict_path(alg, shape):
    """train() on the zero-copy device record == train() on the materialised 11-key dict == oracle."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from test_gpu_learners import build_product
    from golden_cases import case_states
    T, E = 7, 6
    case = ("x", shape, alg, E, T, None, {})
    args, mac, learner_a = build_product(case)
    _, _, learner_b = build_product(case)
    args.epsilon, args.seed = 0.3, 5
    sh = seeded.SHAPES[shape]
    env = SyntheticSMACEnv(E, sh["n_agents"], sh["obs_shape"], sh["state_shape"], sh["n_actions"], T, seed=9)
    ep, _, _, _ = RolloutWorker(env, mac, args).generate_episodes(E)
    d = ep.numpy()
    assert (d["padded"].sum() > 0), "want ragged episodes in this test"
    la = learner_a.train(ep, 0)
    lb = learner_b.train({k: v.copy() for k, v in d.items()}, 0)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    np.testing.assert_allclose(learner_a._flat.grad.cpu().numpy(), learner_b._flat.grad.cpu().numpy(), atol=1e-4, rtol=1e-3)
    _, agent, mixer, v, extra = case_states(case)
    st = learners.LearnerState(args, agent, mixer, v, extra)
    lo, _, _ = learners.train(st, d, 0)
    np.testing.assert_allclose(la, lo, rtol=1e-4)


def test_replay_buffer_device_and_host(golden_dir):
    from marl_amd.common.replaybuffer import ReplayBuffer
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    fix = np.load(golden_dir + "/replay.npz")
    # host mode reproduces the reference ring arithmetic and sampling
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
    # device mode: same ring indices, records gathered on HBM
    T, E = 5, 4
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, buffer_size=10, epsilon=0.2)
    mac, _ = _mac(args)
    env = SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=4)
    w = RolloutWorker(env, mac, args)
    buf = ReplayBuffer(args)
    eps_list = []
    for i in range(3):
        ep, _, _, _ = w.generate_episodes(E)
        eps_list.append(ep.numpy())
        buf.store_episode(ep)
    assert buf.current_size == 10 and buf.current_idx == 2
    np.random.seed(1)
    smp = buf.sample(6).numpy()
    np.random.seed(1)
    idx = np.random.randint(0, 10, 6)
    ring = {k: np.concatenate([eps_list[0][k], eps_list[1][k], eps_list[2][k][:2]], 0) for k in eps_list[0]}
    for k in ring:
        ring[k][0:2] = eps_list[2][k][2:4]     # wrap-around: last two episodes overwrite slots 0,1
        np.testing.assert_array_equal(smp[k], ring[k][idx], err_msg=k)


def test_zero_copy_store_into_replay_ring():
    """record_sink: the rollout kernel writes the episodes straight into the ReplayBuffer's next ring
    slots; the result equals rollout-then-copy, including after the ring wraps (reference
    common/replaybuffer.py:63-80 index arithmetic)."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd.common.replaybuffer import ReplayBuffer
    T, E = 6, 16
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=0.3, seed=3)
    args.buffer_size = 40                      # 16+16 fit, the third store wraps (copy path), the 4th is in place again
    mac, _ = _mac(args)
    wa = RolloutWorker(SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=9), mac, args)
    wb = RolloutWorker(SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=9), mac, args)
    ba, bb = ReplayBuffer(args), ReplayBuffer(args)
    wb.record_sink = bb
    in_place = []
    for it in range(5):
        ea = wa.generate_episodes(E)[0]
        eb = wb.generate_episodes(E)[0]
        in_place.append(getattr(eb.record, "sink_slot", None))
        ba.store_episode(ea)
        bb.store_episode(eb)
        assert (ba.current_idx, ba.current_size) == (bb.current_idx, bb.current_size)
        for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
            n = ba.current_size
            assert torch.equal(getattr(ba.record, f)[:n], getattr(bb.record, f)[:n]), (it, f)
    assert in_place == [0, 16, None, 8, 24]
    # evaluation rollouts never touch the ring
    snap = bb.record.obs.clone()
    ev = wb.generate_episodes(E, evaluate=True)[0]
    assert getattr(ev.record, "sink_slot", None) is None and torch.equal(snap, bb.record.obs)


@pytest.mark.parametrize("alg,shape", [("qmix", "2s3z"), ("qplex", "2s3z"), ("qtran_base", "3s5z")])
def test_replay_sample_is_read_in_place(alg, shape):
    """ReplayBuffer.sample returns a (ring, index) view: the learner reads observations / states in place
    through the episode map (sampling WITH replacement -> duplicates, reference replaybuffer.py:54-60) and the
    result equals training on the gathered copy."""
    from marl_amd.rollout import RolloutWorker, EpisodeBatch
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd.common.replaybuffer import ReplayBuffer
    from test_gpu_learners import build_product
    T, E = 25, 24
    case = ("x", shape, alg, E, T, None, {})
    args, mac, learner_a = build_product(case)
    _, _, learner_b = build_product(case)
    args.epsilon, args.seed, args.buffer_size = 0.3, 5, 64
    sh = seeded.SHAPES[shape]
    env = SyntheticSMACEnv(E, sh["n_agents"], sh["obs_shape"], sh["state_shape"], sh["n_actions"], T, seed=9)
    w = RolloutWorker(env, mac, args)
    buf = ReplayBuffer(args)
    w.record_sink = buf
    for _ in range(2):
        buf.store_episode(w.generate_episodes(E)[0])
    np.random.seed(3)
    batch = buf.sample(40)                                   # 40 draws from 48 stored episodes
    assert batch.ring is buf.record and batch._record is None
    idx = batch.index.cpu().numpy()
    assert len(set(idx.tolist())) < len(idx), "want duplicates"
    la = learner_a.train(batch, 0)
    assert batch._record is None                             # nothing was gathered
    gathered = EpisodeBatch(buf.record.index_select(batch.index))
    lb = learner_b.train(gathered, 0)
    np.testing.assert_allclose(la, lb, rtol=1e-6)
    ga, gb = learner_a._flat.grad.cpu().numpy(), learner_b._flat.grad.cpu().numpy()
    np.testing.assert_allclose(ga, gb, atol=1e-6 * max(1.0, np.abs(gb).max()), rtol=1e-4)
    # generic consumers still get the reference's 11-key dict
    d = batch.numpy()
    assert d["o"].shape == (40, T, sh["n_agents"], sh["obs_shape"])


@pytest.mark.parametrize("eps,evaluate,threads", [(0.0, True, 0), (0.5, False, 0), (0.5, False, 4)])
def test_host_vector_env_matches_device_env(eps, evaluate, threads):
    """R6 for HOST environments: n objects with the reference's serial env API (rollout.py:42,61-64,86-88) behind
    HostVectorEnv give the record of the device synthetic env bit for bit (same lock-step protocol, same batched agent
    step / epsilon-greedy kernels), with ONE H2D and ONE D2H copy per lock-step."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd.env.host_vector import HostVectorEnv
    T, E, env0 = 8, 13, 2
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=eps, seed=77)
    args.anneal_epsilon = 0.01
    mac, agent = _mac(args)
    sy = orl.SynthSMAC(5, 80, 120, 11, T, seed=5)
    henv = HostVectorEnv([orl.SerialSynthEnv(sy, env_id=env0 + i) for i in range(E)], seed=5, env0=env0, n_threads=threads)
    assert henv.get_env_info() == sy.get_env_info()
    wh = RolloutWorker(henv, mac, args)
    wd = RolloutWorker(SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=5, env0=env0), mac, args)
    wd.rollout_mode = "unfused"
    for rollout in range(2):        # the second one continues the episode counters and the epsilon schedule
        eh, rh, winh, sh = wh.generate_episodes(E, evaluate=evaluate)
        ed, rd, wind, sd = wd.generate_episodes(E, evaluate=evaluate)
        for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
            assert torch.equal(getattr(eh.record, f), getattr(ed.record, f)), (rollout, f)
        assert sh == sd and winh == wind and rh == rd
        np.testing.assert_allclose(wh.epsilon, wd.epsilon, rtol=1e-12)
        # one bundle up (+ the first observation) and one action vector down per lock-step
        assert henv.d2h_copies == (rollout + 1) * T and henv.h2d_copies == (rollout + 1) * (T + 1)
    # ... and the oracle's batched restatement of the second rollout
    oep, _, _, osteps, _ = orl.batched_rollout(agent, args, sy, E, eps if evaluate else eps - T * 0.01 if eps > args.min_epsilon else eps,
                                               evaluate=evaluate, rseed=77, env0=env0, episode=1)
    assert sh == osteps
    np.testing.assert_array_equal(eh.numpy()["u"], oep["u"])


def test_host_vector_env_matches_reference_greedy_fixture(golden_dir):
    """the reference RolloutWorker's own greedy record on six serial episodes (tests/golden/rollout.npz: smac_greedy, written
    by the reference's rollout.py) == six host environments stepped in lock-step through the adapter"""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.host_vector import HostVectorEnv
    fix = np.load(golden_dir + "/rollout.npz")
    args = seeded.make_args("2s3z", "qmix", episode_limit=8)
    args.epsilon = 0.0
    mac, _ = _mac(args)
    sy = orl.SynthSMAC(5, 80, 120, 11, 8, seed=5)
    w = RolloutWorker(HostVectorEnv([orl.SerialSynthEnv(sy, env_id=i) for i in range(6)], seed=5), mac, args)
    ep, rew, wins, steps = w.generate_episodes(6, evaluate=True)
    got = ep.numpy()
    for k in ("u", "r", "padded", "terminated", "avail_u", "avail_u_next"):
        np.testing.assert_allclose(got[k], fix["smac_greedy/%s" % k], atol=1e-6, err_msg=k)
    chk = seeded.checksum([got["o"], got["o_next"], got["s"], got["s_next"]])
    np.testing.assert_allclose(chk, float(fix["smac_greedy/o_checksum"]), rtol=1e-6)
    assert steps == int(fix["smac_greedy/steps"]) and list(wins) == list(fix["smac_greedy/wins"])


@pytest.mark.parametrize("shape,E,T,eps", [("2s3z", 37, 8, 0.3), ("2s3z", 1024, 6, 0.5), ("2s3z", 2048, 5, 0.0), ("2s3z", 3000, 5, 1.0),
                                           ("2s3z", 4096, 7, 0.2), ("2s3z", 4500, 4, 0.2), ("3s5z", 300, 6, 0.4), ("3s5z", 2048, 5, 0.1)])
def test_split_rollout_decompositions_agree_bitwise(shape, E, T, eps):
    """csrc/rollout_x6.hip (round 6: the recurrent team runs x W_ih and h W_hh down one accumulator chain, three barriers per lock-step,
    up to five row tiles per workgroup) against csrc/rollout_x6_v1.hip (round 5: gate sums handed over through LDS, four barriers,
    three tiles): the same additions in the same order, so every field of the record and the final hidden state agree BIT FOR BIT -
    at one to five row tiles per workgroup, one and several rounds of workgroups, ragged last workgroups."""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd import experiments
    dims = {"2s3z": (5, 80, 120, 11), "3s5z": (8, 128, 216, 14)}[shape]
    args = seeded.make_args(shape, "qmix", episode_limit=T, epsilon=eps, seed=31)
    args.anneal_epsilon = 0.02
    args.gemm_mode = "bf16x6"
    mac, _ = _mac(args)
    recs = []
    for v1 in (2, 1):           # 2 / 1: force this round's / the round-5 kernel (0 = the library picks by batch size)
        with experiments.override(rollout_v1=v1):
            w = RolloutWorker(SyntheticSMACEnv(E, *dims, T, seed=9, env0=3), mac, args)
            out = []
            for _ in range(2):                  # the second rollout continues the episode counter and the epsilon schedule
                ep, rew, wins, steps = w.generate_episodes(E, evaluate=False)
                out.append((ep.record, mac.hidden_states.clone(), rew, steps, w.epsilon))
            recs.append(out)
    for (ra, ha, rewa, sa, ea), (rb, hb, rewb, sb, eb) in zip(*recs):
        for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
            assert torch.equal(getattr(ra, f), getattr(rb, f)), f
        assert torch.equal(ha, hb) and rewa == rewb and sa == sb and ea == eb
        assert int((ra.u >= 0).sum()) > 0


@pytest.mark.parametrize("N,O,S,A,E,T", [(49, 40, 60, 5, 23, 6), (1, 8, 10, 3, 700, 5), (16, 64, 100, 16, 300, 5), (33, 100, 70, 5, 40, 4),
                                         (10, 176, 322, 18, 37, 6), (10, 176, 322, 18, 1024, 4), (4, 180, 50, 32, 300, 5), (7, 152, 64, 16, 200, 5)])
def test_split_rollout_edge_shapes_match_the_per_step_path(N, O, S, A, E, T):
    """shapes only the round-6 split rollout kernel covers in one launch (an environment of up to 64 agents across several row tiles;
    one-agent environments, 80 to a workgroup; 16 actions = a full DPP row; MMM2-sized agents: seven fc1 chunks and TWO action tiles -
    18 and 32 actions; 175 input columns and 16 actions): the whole-rollout record == the per-step kernels' record
    (fp32 agent step; actions are argmaxes of well separated Q values here) and == the oracle's integer fields"""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from marl_amd import ops
    import types
    assert ops.synth_rollout_x6_supported(N, O, A)
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=0.4, seed=5)
    args.n_agents, args.obs_shape, args.state_shape, args.n_actions = N, O, S, A
    args.anneal_epsilon = 0.03
    args.gemm_mode = "bf16x6"
    mac, agent = _mac(args)
    w = RolloutWorker(SyntheticSMACEnv(E, N, O, S, A, T, seed=4, env0=1), mac, args)
    ep, rew, wins, steps = w.generate_episodes(E)
    w2 = RolloutWorker(SyntheticSMACEnv(E, N, O, S, A, T, seed=4, env0=1), mac, args)
    w2.rollout_mode = "unfused"
    ep2, rew2, wins2, steps2 = w2.generate_episodes(E)
    for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
        assert torch.equal(getattr(ep.record, f), getattr(ep2.record, f)), f
    assert steps == steps2 and wins == wins2
    sy = orl.SynthSMAC(N, O, S, A, T, seed=4)
    oep, orew, owins, osteps, _ = orl.batched_rollout(agent, args, sy, E, 0.4, rseed=5, env0=1)
    np.testing.assert_array_equal(ep.numpy()["u"], oep["u"])
    assert steps == osteps
