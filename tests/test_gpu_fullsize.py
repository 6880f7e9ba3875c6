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
