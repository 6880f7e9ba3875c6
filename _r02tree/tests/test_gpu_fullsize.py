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


@pytest.fixture(scope="module")
def world():
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from test_gpu_learners import build_product
    case = ("full", "2s3z", "qmix", E, T, None, {})
    args, mac, learner = build_product(case)
    args.epsilon, args.anneal_epsilon, args.seed = 0.3, 1e-4, 41
    env = SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=17)
    w = RolloutWorker(env, mac, args)
    ep, rew, wins, steps = w.generate_episodes(E)
    return dict(case=case, args=args, mac=mac, learner=learner, ep=ep, steps=steps, eps_after=w.epsilon)


def test_rollout_partition_invariance_and_oracle_samples(world):
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from golden_cases import case_states
    args, mac, rec = world["args"], world["mac"], world["ep"].record
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


def test_learner_linearity_and_oracle_samples(world):
    from marl_amd.hostutil import DeviceBatch
    from golden_cases import case_states
    learner, rec, args = world["learner"], world["ep"].record, world["args"]
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


def test_qplex_linearity_and_oracle_samples(world):
    """QPLEX (BASELINE config 3 shape) at the full 4096 x 120 batch: the fused lambda-net head kernels walk 30 720
    row tiles per head here.  Same properties as above: the full batch's un-normalised gradient equals the sum over
    its halves, and q_tot / target q_tot of sampled episodes equal the CPU oracle's."""
    from marl_amd.hostutil import DeviceBatch
    from marl_amd.rollout import EpisodeBatch
    from test_gpu_learners import build_product
    from golden_cases import case_states
    case = ("fullq", "2s3z", "qplex", E, T, None, {})
    args, mac, learner = build_product(case)
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


def _shard_world(shape, alg, envs, T, seed, over=None):
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    from test_gpu_learners import build_product
    case = ("shard", shape, alg, envs, T, None, over or {})
    args, mac, learner = build_product(case)
    args.epsilon, args.anneal_epsilon, args.seed = 0.3, 1e-4, seed
    env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, T, seed=seed + 1)
    ep, _, _, _ = RolloutWorker(env, mac, args).generate_episodes(envs)
    return case, args, learner, ep.record


def _sub_batch_vs_oracle(case, args, learner, rec, idx, Tm, name, tol=1e-4):
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
        parity.close(name, "grad " + pn, g, og.detach().numpy(), tol=tol)


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


def test_config4_qtran_3s5z_shard_fullsize():
    """BASELINE config 4 at its per-GPU shard (QTRAN-base, 3s5z shape, 2048 envs / 4 GPUs = 512 envs x T = 150):
    614 400 agent rows through the fused joint-Q / V head kernels and the hidden-state-gradient BPTT variant.
    (a) the un-normalised [gradients | three loss numerators | sum(mask)] of the shard equal the sum over its halves;
    (b) on sampled episodes - same T = 150 loop - joint_q, target joint_q, v, joint_q_hat, the three loss numerators
    and every parameter gradient equal the CPU oracle's within 1e-4 of their scale."""
    from marl_amd.hostutil import DeviceBatch
    E4, T4 = 512, 150
    case, args, learner, rec = _shard_world("3s5z", "qtran_base", E4, T4, seed=23)
    assert int(rec.padded.sum().item()) > 0
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T4
    _linearity(learner, rec, Tm, E4, 4, "full:cfg4_qtran_3s5z_512x150")
    _sub_batch_vs_oracle(case, args, learner, rec, [0, 1, 255, 256, 300, 511], Tm, "full:cfg4_qtran_3s5z_512x150")


def test_config5_qmix_mmm2_shard_fullsize():
    """BASELINE config 5 at its per-GPU shard (QMIX, MMM2 shape, 8192 envs / 8 GPUs = 1024 envs x T = 120, fp32):
    S = 322 states (rows not 16-byte aligned in a dense layout), 10 agents, two action tiles in the agent kernels.
    Same two properties as config 4: shard linearity and oracle parity (forward, loss, all gradients) on samples."""
    from marl_amd.hostutil import DeviceBatch
    E5, T5 = 1024, 120
    case, args, learner, rec = _shard_world("MMM2", "qmix", E5, T5, seed=29)
    Tm = DeviceBatch.first_terminated_len(rec.term, args.episode_limit)
    assert Tm == T5
    _linearity(learner, rec, Tm, E5, 2, "full:cfg5_qmix_MMM2_1024x120")
    _sub_batch_vs_oracle(case, args, learner, rec, [0, 1, 511, 512, 700, 1023], Tm, "full:cfg5_qmix_MMM2_1024x120")
