"""BASELINE-size parity (QMIX, 2s3z shape, 4096 envs, T=120 - the bench workload) through size-independent
properties, plus direct oracle comparisons on sampled environments / episodes:
  * rollout: the 4096-env record is bit-identical to two 2048-env rollouts glued together (independence of the
    workgroup decomposition) and to the CPU oracle on sampled environments;
  * learner: the un-normalised gradient / loss numerators of the full batch equal the sum over its two halves
    (linearity of the data-parallel reduction), and the forward quantities of sampled episodes equal
    the oracle's within 1e-4."""
import numpy as np
import pytest
import torch

from oracle import seeded, rollout as orl, learners

pytestmark = pytest.mark.gpu

E, T = 4096, 120


MODES = ["f32", "bf16x6"]       # both arithmetic modes of the dense products, same bounds (tests/conftest.py: gemm_mode)


@pytest.fixture(scope="module")
def world():
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from test_gpu_learners import build_product
    case = ("full", "2s3z", "qmix", E, T, None, {})
    args, mac, learner = build_product(case, "f32")
    args.epsilon, args.anneal_epsilon, args.seed = 0.3, 1e-4, 41
    env = SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=17)
    w = RolloutWorker(env, mac, args)
    ep, rew, wins, steps = w.generate_episodes(E)
    # the same weights behind a learner in split arithmetic (the record above is shared: the rollout's integer fields do not
    # depend on the mode - test_gpu_rollout.py checks that - and the learner tests below read the record only)
    _, _, learner6 = build_product(case, "bf16x6")
    return dict(case=case, args=args, mac=mac, learner=learner, learners={"f32": learner, "bf16x6": learner6}, ep=ep, steps=steps,
                eps_after=w.epsilon)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_rollout_partition_invariance_and_oracle_samples(world, gemm_mode):
    """(bf16x6: the 4096-env rollout on csrc/rollout_x6.hip - 512 workgroups of 8 environments in two rounds; its halves run 256
    workgroups each)"""
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from golden_cases import case_states
    import copy
    args, mac, rec = world["args"], world["mac"], world["ep"].record
    if gemm_mode == "bf16x6":
        args = copy.copy(args)
        args.gemm_mode = "bf16x6"
        w6 = RolloutWorker(SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=17), mac, args)
        w6.epsilon = 0.3
        ep6, _, _, steps6 = w6.generate_episodes(E)
        rec = ep6.record
        assert steps6 == world["steps"]
        # the environment does not depend on the arithmetic; the chosen actions agree with the fp32 kernel's except in episodes where
        # two available actions' Q values came within rounding of each other (1.7 M greedy choices; after such a choice the episode's
        # later actions differ too): a fraction of a percent of the episodes at most
        for f in ("obs", "state", "avail", "term", "padded", "length", "won"):
            assert torch.equal(getattr(rec, f), getattr(world["ep"].record, f)), f
        n_diff = int((rec.u != world["ep"].record.u).flatten(1).any(1).sum().item())
        print("episodes whose actions differ between the fp32 and the bf16x6 rollout: %d of %d" % (n_diff, E))
        assert n_diff <= E // 100
    assert world["steps"] == int(rec.length.sum().item())
    assert int(rec.padded.sum().item()) > 0, "ragged episodes wanted"
    half = E // 2
    for part in range(2):
        env = SyntheticSMACEnv(half, 5, 80, 120, 11, T, seed=17, env0=part * half)
        w = RolloutWorker(env, mac, args)
        w.epsilon = 0.3
        ep, _, _, _ = w.generate_episodes(half)
        for f in ("obs", "state", "avail", "u", "r", "term", "padded", "length", "won"):
            assert torch.equal(getattr(ep.record, f), getattr(rec, f)[part * half:(part + 1) * half]), (part, f)
    # CPU oracle on sampled environments (bit-exact integer fields, 1e-6 floats)
    _, agent, _, _, _ = case_states(world["case"])
    sy = orl.SynthSMAC(5, 80, 120, 11, T, seed=17)
    from marl_amd.rollout import EpisodeBatch
    for env0 in (0, 1023, 2048, 4090):
        oep, _, _, _, _ = orl.batched_rollout(agent, args, sy, 3, 0.3, rseed=41, env0=env0)
        got = EpisodeBatch(rec.slice(env0, env0 + 3)).numpy()
        for k in ("u", "padded", "terminated", "avail_u", "avail_u_next", "u_onehot"):
            np.testing.assert_array_equal(got[k], np.asarray(oep[k], dtype=got[k].dtype), err_msg="%s env0=%d" % (k, env0))
        for k in ("o", "o_next", "s", "s_next", "r"):
            np.testing.assert_allclose(got[k], oep[k], atol=1e-6, err_msg=k)


def _grads(learner, rec, Tfix):
    from marl_amd.hostutil import DeviceBatch
    db = DeviceBatch.from_record(rec, learner.args, T=Tfix)
    learner._forward_backward(db)
    torch.cuda.synchronize()
    return learner._flat.gradx.detach().cpu().double().numpy().copy(), {k: v.detach().cpu().numpy().copy() for k, v in learner._dbg.items()}


@pytest.mark.parametrize("gemm_mode", MODES)
def test_learner_linearity_and_oracle_samples(world, gemm_mode):
    from marl_amd.hostutil import DeviceBatch
    from golden_cases import case_states
    learner, rec = world["learners"][gemm_mode], world["ep"].record
    args = learner.args
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    full, dbg = _grads(learner, rec, Tm)
    half = E // 2
    ga, _ = _grads(learner, rec.slice(0, half), Tm)
    gb, _ = _grads(learner, rec.slice(half, E), Tm)
    # [gradients | loss numerator | sum(mask)]: sums over disjoint episode shards (exactness rule of SURVEY 8e)
    tot = ga + gb
    n = learner._flat.n                                   # gradx = [grads (n) | loss numerator | sum(mask) | - | -]
    assert tot[n + 1] == full[n + 1] and full[n + 1] == float((1.0 - rec.padded[:, :Tm]).sum().item())
    np.testing.assert_allclose(full[n], tot[n], rtol=2e-5)
    scale = np.abs(full[:n]).max()
    np.testing.assert_allclose(full[:n] / scale, tot[:n] / scale, atol=2e-5)
    # forward quantities of sampled episodes vs the CPU oracle
    _, agent, mixer, _, _ = case_states(world["case"])
    st = learners.LearnerState(args, agent, mixer)
    idx = [0, 1, 2047, 2048, 3000, 4095]
    from marl_amd.rollout import EpisodeBatch
    sub = EpisodeBatch(rec.index_select(torch.as_tensor(idx, device=rec.obs.device))).numpy()
    _, inter = learners.q_forward(st, sub, T=Tm)
    np.testing.assert_allclose(dbg["q_evals"][idx], inter["q_evals"].detach().numpy(), atol=1e-4)
    qt_o = inter["q_targets"].detach().numpy()            # the oracle's copy carries the -9999999 availability mask
    ok = qt_o > -1e6
    np.testing.assert_allclose(dbg["q_targets"][idx][ok], qt_o[ok], atol=1e-4)
    # mixer outputs on the unpadded steps (at the first padded step the de-duplicated record still shows the
    # terminal state where the reference dict has zeros; that step is masked out of the loss, q_learner.py:166)
    live = sub["padded"][:, :Tm, 0] == 0
    for k in ("q_tot", "q_tot_target"):
        a_ = dbg[k].reshape(E, Tm)[idx]
        b_ = inter[k].detach().numpy().reshape(len(idx), Tm)
        np.testing.assert_allclose(a_[live], b_[live], atol=2e-4, err_msg=k)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_qplex_linearity_and_oracle_samples(world, gemm_mode):
    """QPLEX (BASELINE config 3 shape) at the full 4096 x 120 batch: the fused lambda-net head kernels walk 30 720
    row tiles per head here.  Same properties as above: the full batch's un-normalised gradient equals the sum over
    its halves, and q_tot / target q_tot of sampled episodes equal the CPU oracle's."""
    from marl_amd.hostutil import DeviceBatch
    from marl_amd.rollout import EpisodeBatch
    from test_gpu_learners import build_product
    from golden_cases import case_states
    case = ("fullq", "2s3z", "qplex", E, T, None, {})
    args, mac, learner = build_product(case, gemm_mode)
    rec = world["ep"].record
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    full, dbg = _grads(learner, rec, Tm)
    half = E // 2
    ga, _ = _grads(learner, rec.slice(0, half), Tm)
    gb, _ = _grads(learner, rec.slice(half, E), Tm)
    tot = ga + gb
    n = learner._flat.n
    assert tot[n + 1] == full[n + 1]
    np.testing.assert_allclose(full[n], tot[n], rtol=5e-5)
    scale = np.abs(full[:n]).max()
    np.testing.assert_allclose(full[:n] / scale, tot[:n] / scale, atol=5e-5)
    _, agent, mixer, _, _ = case_states(case)
    st = learners.LearnerState(args, agent, mixer)
    idx = [0, 1, 2047, 2048, 4095]
    sub = EpisodeBatch(rec.index_select(torch.as_tensor(idx, device=rec.obs.device))).numpy()
    _, inter = learners.q_forward(st, sub, T=Tm)
    live = sub["padded"][:, :Tm, 0] == 0
    for k in ("q_tot", "q_tot_target"):
        a_ = dbg[k].reshape(E, Tm)[idx]
        b_ = inter[k].detach().numpy().reshape(len(idx), Tm)
        np.testing.assert_allclose(a_[live], b_[live], atol=3e-4, rtol=1e-4, err_msg=k)


def _shard_world(shape, alg, envs, T, seed, over=None, gemm_mode=None):
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from test_gpu_learners import build_product
    case = ("shard", shape, alg, envs, T, None, over or {})
    args, mac, learner = build_product(case, gemm_mode)
    args.epsilon, args.anneal_epsilon, args.seed = 0.3, 1e-4, seed
    env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, T, seed=seed + 1)
    ep, _, _, _ = RolloutWorker(env, mac, args).generate_episodes(envs)
    return case, args, learner, ep.record


def _sub_batch_vs_oracle(case, args, learner, rec, idx, Tm, name, tol=1e-4, grad_tol=None):
    """The product on the sampled episodes alone (full T loop: same trip count, BPTT variant and dispatch as the full
    batch) vs the CPU oracle on the same episodes: forward tensors, loss numerators and EVERY parameter gradient."""
    import parity
    from marl_amd.rollout import EpisodeBatch
    from golden_cases import case_states
    from test_gpu_learners import named_product_params
    sub_rec = rec.index_select(torch.as_tensor(idx, device=rec.obs.device))
    g_sub, dbg = _grads(learner, sub_rec, Tm)
    sub = EpisodeBatch(sub_rec).numpy()
    _, agent, mixer, v, extra = case_states(case)
    st = learners.LearnerState(args, agent, mixer, v, extra)
    live = sub["padded"][:, :Tm, 0] == 0
    n = learner._flat.n
    if args.alg.startswith("qtran"):
        loss, inter = learners.qtran_forward(st, sub, T=Tm)
        den = float(inter["den"])
        for k, ok in (("joint_q", "joint_q_evals"), ("joint_q_targets", "joint_q_targets"), ("v", "v"),
                      ("joint_q_hat", "joint_q_hat_opt")):
            a_ = dbg[k].reshape(len(idx), Tm)
            b_ = inter[ok].detach().numpy().reshape(len(idx), Tm)
            parity.close(name, "sampled " + k, a_[live], b_[live], tol=tol)
        want = [float(inter[k]) * den for k in ("l_td", "l_opt", "l_nopt")] + [den]
        parity.close(name, "loss numerators", g_sub[n:n + 4], np.array(want), tol=tol)
    else:
        loss, inter = learners.q_forward(st, sub, T=Tm)
        den = float(inter["den"])
        for k in ("q_tot", "q_tot_target"):
            a_ = dbg[k].reshape(len(idx), Tm)
            b_ = inter[k].detach().numpy().reshape(len(idx), Tm)
            parity.close(name, "sampled " + k, a_[live], b_[live], tol=tol)
        parity.close(name, "loss numerator", g_sub[n:n + 2], np.array([float(inter["num"]), den]), tol=tol)
    parity.close(name, "sampled q_evals", dbg["q_evals"], inter["q_evals"].detach().numpy(), tol=tol)
    ograds = learners._grads(st, loss)
    for (pn, p) in named_product_params(learner):
        og = ograds.get(pn)
        g = p.grad.detach().cpu().numpy() / den
        if og is None:
            assert np.all(g == 0), pn
            continue
        parity.close(name, "grad " + pn, g, og.detach().numpy(), tol=grad_tol(pn) if grad_tol else tol)


def _linearity(learner, rec, Tm, E_, nstats, name, tol=5e-5):
    import parity
    full, _ = _grads(learner, rec, Tm)
    half = E_ // 2
    ga, _ = _grads(learner, rec.slice(0, half), Tm)
    gb, _ = _grads(learner, rec.slice(half, E_), Tm)
    tot = ga + gb
    n = learner._flat.n
    assert tot[n + nstats - 1] == full[n + nstats - 1] == float((1.0 - rec.padded[:, :Tm]).sum().item())
    parity.close(name, "loss numerators full vs halves", full[n:n + nstats - 1], tot[n:n + nstats - 1], tol=tol)
    parity.close(name, "gradient full vs halves", full[:n], tot[:n], tol=tol)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_config4_qtran_3s5z_shard_fullsize(gemm_mode):
    """BASELINE config 4 at its per-GPU shard (QTRAN-base, 3s5z shape, 2048 envs / 4 GPUs = 512 envs x T = 150):
    614 400 agent rows through the fused joint-Q / V head kernels and the hidden-state-gradient BPTT variant.
    (a) the un-normalised [gradients | three loss numerators | sum(mask)] of the shard equal the sum over its halves;
    (b) on sampled episodes - same T = 150 loop - joint_q, target joint_q, v, joint_q_hat, the three loss numerators
    and every parameter gradient equal the CPU oracle's within 1e-4 of their scale."""
    from marl_amd.hostutil import DeviceBatch
    E4, T4 = 512, 150
    case, args, learner, rec = _shard_world("3s5z", "qtran_base", E4, T4, seed=23, gemm_mode=gemm_mode)
    assert int(rec.padded.sum().item()) > 0
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T4
    name = "full:cfg4_qtran_3s5z_512x150[%s]" % gemm_mode
    _linearity(learner, rec, Tm, E4, 4, name)
    _sub_batch_vs_oracle(case, args, learner, rec, [0, 1, 255, 256, 300, 511], Tm, name)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_config5_qmix_mmm2_shard_fullsize(gemm_mode):
    """BASELINE config 5 at its per-GPU shard (QMIX, MMM2 shape, 8192 envs / 8 GPUs = 1024 envs x T = 120, fp32):
    S = 322 states (rows not 16-byte aligned in a dense layout), 10 agents, two action tiles in the agent kernels.
    Same two properties as config 4: shard linearity and oracle parity (forward, loss, all gradients) on samples."""
    from marl_amd.hostutil import DeviceBatch
    E5, T5 = 1024, 120
    case, args, learner, rec = _shard_world("MMM2", "qmix", E5, T5, seed=29, gemm_mode=gemm_mode)
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T5
    name = "full:cfg5_qmix_MMM2_1024x120[%s]" % gemm_mode
    _linearity(learner, rec, Tm, E5, 2, name)
    _sub_batch_vs_oracle(case, args, learner, rec, [0, 1, 511, 512, 700, 1023], Tm, name)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_config5_qmix_mmm2_bf16_mixer_learner_vs_oracle(gemm_mode):
    """BASELINE config 5 as it is quoted ("bf16 mixer with MFMA") at its per-GPU shard, learner level: QMIX on MMM2, 1024 envs x
    T = 120, args.mixer_dtype = "bf16".  The reference has no such mode; the oracle restates it (oracle/nets.py:_LinBf16 - both
    operands of the four state-conditioned hypernet GEMMs rounded to bf16, fp32 accumulation, forward AND weight gradient), which
    makes it an EXTERNAL check of the reduced-precision path at the north-star tolerance: products of bf16 values
    are exact in fp32, only the accumulation order differs.  (a) what the FULL-batch launches produced (122 880 rows: the
    resident-weights bf16 forward for the target mixer, the streaming bf16 kernel with the folded loss for the eval mixer) for
    sampled episodes: q_evals, q_targets, q_tot, q_tot_target at 1e-4 of scale; (b) the sampled sub-batch with the loss numerator and
    every parameter gradient (1e-4; the four bf16 weight-gradient GEMMs at 5e-3 - measured 2e-3); (c) shard linearity."""
    from marl_amd.hostutil import DeviceBatch
    from marl_amd import ops
    E5, T5 = 1024, 120
    case, args, learner, rec = _shard_world("MMM2", "qmix", E5, T5, seed=29, over={"mixer_dtype": "bf16"}, gemm_mode=gemm_mode)
    assert learner.mixer._bf16() and args.mixer_dtype == "bf16"
    assert ops.qmix_wide_fwd_kernel(E5 * T5, args.n_agents, args.state_shape, bf16=True) == "qmix_wide_res_fwd_kernel"
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T5
    name = "full:cfg5_qmix_MMM2_1024x120_bf16mixer[%s]" % gemm_mode
    idx = [0, 1, 511, 512, 700, 1023]
    _, dbg = _grads(learner, rec, Tm)
    _full_batch_samples_vs_oracle(case, args, dbg, rec, idx, Tm, name)
    # the weight gradients of the four bf16 GEMMs round dhy and the states to bf16 before multiplying: a 1e-7 difference in dhy
    # between two correct evaluations can move an element across a bf16 rounding boundary (2^-8 of that term), so these four
    # tensors are compared at 5e-3 (measured: 2e-3); everything else - the loss, dq and
    # with it every agent gradient, the biases, hyper_b2.2 - at 1e-4.  (The bf16 weight-gradient GEMM is its own flag bit of the C-ABI;
    # args.mixer_wgrad_dtype = "fp32" keeps that GEMM on fp32 MFMAs: tests/test_gpu_kernels.py::test_qmix_wide[...-fwd].)
    bf_w = ("mixer.hyper_w1.weight", "mixer.hyper_b1.weight", "mixer.hyper_w2.weight", "mixer.hyper_b2.0.weight")
    _sub_batch_vs_oracle(case, args, learner, rec, idx, Tm, name, grad_tol=lambda pn: 5e-3 if pn in bf_w else 1e-4)
    _linearity(learner, rec, Tm, E5, 2, name)


def _full_batch_samples_vs_oracle(case, args, dbg, rec, idx, Tm, name, tol=1e-4):
    """Forward tensors the FULL-batch launch produced (whatever schedule and kernel variants its size selected) for the
    sampled episodes vs the CPU oracle run on those episodes alone (rows are independent)."""
    import parity
    from marl_amd.rollout import EpisodeBatch
    from golden_cases import case_states
    _, agent, mixer, v, extra = case_states(case)
    st = learners.LearnerState(args, agent, mixer, v, extra)
    sub = EpisodeBatch(rec.index_select(torch.as_tensor(idx, device=rec.obs.device))).numpy()
    _, inter = learners.q_forward(st, sub, T=Tm)
    E_ = rec.E
    parity.close(name, "full-batch q_evals", dbg["q_evals"][idx], inter["q_evals"].detach().numpy(), tol=tol)
    qt_o = inter["q_targets"].detach().numpy()              # the oracle's copy carries the -9999999 availability mask
    ok = qt_o > -1e6
    parity.close(name, "full-batch q_targets", dbg["q_targets"][idx][ok], qt_o[ok], tol=tol)
    live = sub["padded"][:, :Tm, 0] == 0
    for k in ("q_tot", "q_tot_target"):
        a_ = dbg[k].reshape(E_, Tm)[idx]
        b_ = inter[k].detach().numpy().reshape(len(idx), Tm)
        parity.close(name, "full-batch " + k, a_[live], b_[live], tol=tol)


def _chain_schedule_case(alg, envs, name, gemm_mode):
    """A shard whose size selects PairedUnroll.run_chain (eval current-Q -> double-Q continuation on 160 CUs, target
    unroll beside it on 96 CUs of a side stream; reference q_learner.py:96-117, quirk Q1: the continuation starts from the
    eval pass's final hidden state and here also READS its input-side gate sums across the stream fork)."""
    from marl_amd.hostutil import DeviceBatch
    T2 = 120
    name = "%s[%s]" % (name, gemm_mode)
    case, args, learner, rec = _shard_world("2s3z", alg, envs, T2, seed=31, gemm_mode=gemm_mode)
    assert int(rec.padded.sum().item()) > 0, "ragged episodes wanted"
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T2
    split = learner.pair.chain_split(envs * args.n_agents, Tm, args.obs_shape)
    if split is None and gemm_mode == "bf16x6":
        # the split kernels' step-time model may prefer the plain schedule at this size (they run rounds of one- / two-tile
        # workgroups): the chain schedule is then forced, so that it stays under full-size parity in this mode too
        learner.pair.forced_split = 160
        split = learner.pair.chain_split(envs * args.n_agents, Tm, args.obs_shape)
    assert split is not None and split[0] + split[1] == 256, split
    # (a) the chain schedule vs the plain schedule (pair, then the continuation over the whole chip) on the same record:
    # rows are independent and every kernel variant accumulates a row's products in the same order -> bitwise equal
    g_chain, dbg_chain = _grads(learner, rec, Tm)
    learner.pair.chain, forced = False, learner.pair.forced_split
    learner.pair.forced_split = None
    assert learner.pair.chain_split(envs * args.n_agents, Tm, args.obs_shape) is None
    g_plain, dbg_plain = _grads(learner, rec, Tm)
    learner.pair.chain, learner.pair.forced_split = True, forced
    for k in ("q_evals", "q_targets", "q_tot", "q_tot_target"):
        np.testing.assert_array_equal(dbg_chain[k], dbg_plain[k], err_msg=name + " chain vs plain " + k)
    np.testing.assert_array_equal(g_chain, g_plain, err_msg=name + " chain vs plain gradx")
    # (b) what the chain launches produced for sampled episodes vs the CPU oracle
    idx = [0, 1, envs // 2 - 1, envs // 2, envs - 200, envs - 1]
    _full_batch_samples_vs_oracle(case, args, dbg_chain, rec, idx, Tm, name)
    # (c) shard linearity (the full shard runs the chain schedule; its halves whatever their size selects) and the sampled
    # sub-batch incl. every gradient
    _linearity(learner, rec, Tm, envs, 2, name)
    _sub_batch_vs_oracle(case, args, learner, rec, idx, Tm, name)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_config2_qmix_2s3z_1024_chain_schedule(gemm_mode):
    """BASELINE config 2 (QMIX, 2s3z, 1024 envs x T = 120 on one MI355X): its size selects the chain schedule."""
    _chain_schedule_case("qmix", 1024, "full:cfg2_qmix_2s3z_1024x120", gemm_mode)


@pytest.mark.parametrize("gemm_mode", MODES)
def test_config3_qplex_2s3z_512_shard_chain_schedule(gemm_mode):
    """BASELINE config 3 at its per-GPU shard (QPLEX, 2s3z, 4096 envs / 8 GPUs = 512 envs x T = 120)."""
    _chain_schedule_case("qplex", 512, "full:cfg3_qplex_2s3z_512x120", gemm_mode)


def test_qplex_mmm2_heads_run_fused_and_match_oracle():
    """QPLEX on an MMM2-sized map (off the five BASELINE configurations): state 322 (not a multiple of 4) and
    [state | 10 x 18 one-hot actions] = 502 input columns - the lambda-net families (mixer.py:117-145) and the transformation
    pair take the fused head kernels' K1 > 192 variants (kept activations, x^T staged in two passes), not the marl_linear
    composition.  Shard linearity + every gradient vs the CPU oracle on sampled episodes."""
    from marl_amd import ops
    from marl_amd.hostutil import DeviceBatch
    Eq, Tq = 96, 40
    case, args, learner, rec = _shard_world("MMM2", "qplex", Eq, Tq, seed=37)
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == Tq
    mx = learner.mixer
    rows = Eq * Tm
    s = DeviceBatch.from_record(rec, args, T=Tm).s               # what the learner hands the mixer (rows of the (T+1)-slot storage)
    xs = ops.src(s)
    xsa = ops.src(s, idx=torch.zeros(rows, args.n_agents, dtype=torch.int32, device=rec.obs.device), nhot=args.n_agents,
                  hot_w=args.n_actions)
    from marl_amd import experiments
    if experiments.get("mlp3_keep") != 0:          # (the A/B switch sends these shapes to the marl_linear composition)
        assert mx._fused_transform(xs) is not None and ops.mlp3_needs_kept(xs, args.state_shape)
        for fname, mods, nout in mx.si_weight.families():
            assert mx._fused_family(mods, xsa if fname == "ac" else xs, nout) is not None, fname
    _linearity(learner, rec, Tm, Eq, 2, "full:qplex_MMM2_96x40")
    _sub_batch_vs_oracle(case, args, learner, rec, [0, 1, 47, 95], Tm, "full:qplex_MMM2_96x40")
