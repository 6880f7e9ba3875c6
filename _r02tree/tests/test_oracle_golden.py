"""Pins the CPU oracle against golden vectors captured from the real reference
(tests/golden/make_golden.py).  CPU only - runs under ``-m "not gpu"``."""
import os

import numpy as np
import pytest
import torch

from oracle import seeded, nets, learners, rollout as orl

from golden_cases import CASES, TRAIN_STEPS, load_fixture, build_oracle_state

torch.set_num_threads(2)


def _pinned(fix, prefix):
    names = sorted({k[len(prefix) + 1:].rsplit("/", 1)[0] for k in fix.files if k.startswith(prefix + "/")})
    return names


def check_pins(fix, prefix, named, atol, rtol):
    """Compare tensors against norm/sum/sample pins written by make_golden.pin()."""
    names = _pinned(fix, prefix)
    assert names, prefix
    got = dict(named)
    for n in names:
        if "%s/%s/none" % (prefix, n) in fix.files:
            assert got.get(n) is None, n
            continue
        a = got[n].detach().cpu().numpy().astype(np.float64).ravel()
        samp = fix["%s/%s/samp" % (prefix, n)]
        np.testing.assert_allclose(a[seeded.sample_indices(a.size)], samp, atol=atol, rtol=rtol, err_msg=prefix + n)
        np.testing.assert_allclose(np.sqrt((a * a).sum()), fix["%s/%s/norm" % (prefix, n)], atol=atol * 10, rtol=rtol,
                                   err_msg=prefix + n)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_pieces(case, golden_dir):
    name, shape, alg, B, T, lengths, over = case
    fix = load_fixture(golden_dir, name)
    args, st = build_oracle_state(case)
    batch = seeded.make_batch(args, B, seed=100, lengths=lengths)
    assert abs(seeded.checksum(batch) - float(fix["meta/batch_checksum"])) < 1e-6
    bt = learners.to_tensors(batch, T)
    N, H = args.n_agents, args.rnn_hidden_dim
    with torch.no_grad():
        h0 = torch.zeros(B * N, H)
        q_cur, h_cur, h_last = nets.agent_unroll(st.agent, bt["o"], nets.shifted_onehot(bt["u_onehot"]), h0)
        q_cont, _, _ = nets.agent_unroll(st.agent, bt["o_next"], bt["u_onehot"], h_last)
        q_nxt, h_nxt, _ = nets.agent_unroll(st.agent, bt["o_next"], bt["u_onehot"], h0)
        tol = dict(atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(q_cur.numpy(), fix["fwd/q_cur"], **tol)
        np.testing.assert_allclose(h_cur.numpy(), fix["fwd/h_cur"], **tol)
        np.testing.assert_allclose(q_nxt.numpy(), fix["fwd/q_next"], **tol)
        np.testing.assert_allclose(h_nxt.numpy(), fix["fwd/h_next"], **tol)
        np.testing.assert_allclose(q_cont.numpy(), fix["fwd/q_next_cont"], **tol)
        qc = torch.gather(q_cur, 3, bt["u"]).squeeze(3)
        if alg == "vdn":
            np.testing.assert_allclose(nets.vdn(qc).numpy(), fix["fwd/q_tot"], **tol)
        elif alg == "qmix":
            np.testing.assert_allclose(nets.qmix(st.mixer, qc, bt["s"], args).numpy(), fix["fwd/q_tot"],
                                       atol=1e-4, rtol=1e-5)
        elif alg == "qplex":
            qd = q_cur.clone(); qd[bt["avail_u"] == 0] = learners.MASK_BIG
            mx = qd.max(dim=3)[0]
            np.testing.assert_allclose(nets.qplex(st.mixer, qc, bt["s"], args, is_v=True).numpy(), fix["fwd/v_tot"], **tol)
            np.testing.assert_allclose(nets.qplex(st.mixer, qc, bt["s"], args, actions=bt["u_onehot"], max_q_i=mx).numpy(),
                                       fix["fwd/a_tot"], **tol)
            np.testing.assert_allclose(nets.qplex_lambda(st.mixer, bt["s"], bt["u_onehot"], args).numpy(),
                                       fix["fwd/lambda"], **tol)
        else:
            np.testing.assert_allclose(nets.qtran_q(st.mixer, bt["s"], h_cur, bt["u_onehot"], args).numpy(),
                                       fix["fwd/joint_q"], **tol)
            np.testing.assert_allclose(nets.qtran_v(st.v, bt["s"], h_cur, args).numpy(), fix["fwd/v"], **tol)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_train_steps(case, golden_dir):
    """loss, pre-clip grads, grad norm and post-step params over 4 updates incl. target sync."""
    name, shape, alg, B, T, lengths, over = case
    fix = load_fixture(golden_dir, name)
    args, st = build_oracle_state(case)
    for i, ts in enumerate(TRAIN_STEPS):
        batch = seeded.make_batch(args, B, seed=100 + i, lengths=lengths)
        loss, grads, inter = learners.train(st, batch, ts)
        # later steps amplify fp32 rounding through RMSprop's 1/sqrt(v): loosen progressively
        rt = 2e-5 * (10 ** i)
        np.testing.assert_allclose(loss, fix["losses"][i], rtol=rt, atol=1e-6, err_msg="loss step %d" % i)
        np.testing.assert_allclose(inter["grad_norm"], float(fix["step%d/grad_norm" % i]), rtol=rt * 5)
        if i <= 1:
            check_pins(fix, "step%d/grad" % i, list(grads.items()), atol=2e-5 * (1 + 50 * i), rtol=1e-3 * (1 + 10 * i))
            check_pins(fix, "step%d/param" % i, st.named_params(), atol=2e-5 * (1 + 50 * i), rtol=1e-4)
        check_pins(fix, "step%d/target_agent" % i, [("agent." + k, v) for k, v in st.target_agent.items()],
                   atol=1e-3, rtol=1e-3)
    assert inter["T"] == int(fix["meta/T_used"])


def test_matrix_rollout_matches_reference(golden_dir):
    """serial_rollout restatement vs RolloutWorker on the matrix game (config 1, 32 episodes)."""
    fix = np.load(os.path.join(golden_dir, "rollout.npz"))
    args = seeded.make_args("matrix", "vdn")
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11)
    env = orl.MatrixGame([[8, -12, -12], [-12, 0, 0], [-12, 0, 0]])
    for tag, eps in (("eps1", 1.0), ("eps03", 0.3)):
        np.random.seed(7)
        ep, rew, wins, steps, eps_after = orl.serial_rollout(agent, args, env, 32, eps)
        for k, v in ep.items():
            np.testing.assert_allclose(np.asarray(v, dtype=np.float64), fix["matrix_%s/%s" % (tag, k)], atol=1e-6, err_msg=k)
        assert steps == int(fix["matrix_%s/steps" % tag])
        np.testing.assert_allclose(eps_after, float(fix["matrix_%s/eps_after" % tag]), rtol=1e-12)
        np.testing.assert_allclose(rew, fix["matrix_%s/rewards" % tag])
    ge = env.get_episodes()
    for k, v in ge.items():
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), fix["matrix_get_episodes/" + k])


def test_smac_shaped_rollout_matches_reference(golden_dir):
    """serial + batched restatements vs the reference RolloutWorker on the synthetic env."""
    fix = np.load(os.path.join(golden_dir, "rollout.npz"))
    args = seeded.make_args("2s3z", "qmix", episode_limit=8)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11, scale=3.0)
    for tag, eps, evaluate in (("greedy", 0.0, True), ("eps05", 0.5, False)):
        sy = orl.SynthSMAC(5, 80, 120, 11, 8, seed=5)
        np.random.seed(9)
        args.epsilon = eps
        ep, rew, wins, steps, eps_after = orl.serial_rollout(agent, args, orl.SerialSynthEnv(sy), 6, eps, evaluate)
        for k in ("u", "r", "padded", "terminated", "avail_u", "avail_u_next"):
            np.testing.assert_allclose(np.asarray(ep[k], dtype=np.float64), fix["smac_%s/%s" % (tag, k)], atol=1e-6, err_msg=k)
        chk = seeded.checksum([ep["o"], ep["o_next"], ep["s"], ep["s_next"]])
        np.testing.assert_allclose(chk, float(fix["smac_%s/o_checksum" % tag]), rtol=1e-9)
        assert steps == int(fix["smac_%s/steps" % tag])
        assert list(wins) == list(fix["smac_%s/wins" % tag])
        np.testing.assert_allclose(eps_after, float(fix["smac_%s/eps_after" % tag]), rtol=1e-12)
    # batched restatement == reference serial run when greedy (no RNG involved)
    epb, rewb, winsb, stepsb, _ = orl.batched_rollout(agent, args, sy, 6, 0.0, evaluate=True)
    for k in ("u", "r", "padded", "terminated", "avail_u", "avail_u_next"):
        np.testing.assert_allclose(np.asarray(epb[k], dtype=np.float64), fix["smac_greedy/%s" % k], atol=1e-6, err_msg=k)
    chk = seeded.checksum([epb["o"].astype(np.float64), epb["o_next"].astype(np.float64),
                           epb["s"].astype(np.float64), epb["s_next"].astype(np.float64)])
    np.testing.assert_allclose(chk, float(fix["smac_greedy/o_checksum"]), rtol=1e-9)
    assert stepsb == int(fix["smac_greedy/steps"])


def test_max_episode_len_quirk_q2():
    term = np.zeros((2, 6, 1)); term[1, 2, 0] = 1; term[1, 3:, 0] = 1
    assert learners.max_episode_len(term, 6) == 3          # unterminated episode ignored
    assert learners.max_episode_len(np.zeros((2, 6, 1)), 6) == 6


@pytest.mark.parametrize("alg", ["vdn", "qplex", "qtran_base"])
def test_oracle_on_reference_checkpoints(alg, golden_dir):
    """The oracle fed with the state dicts the reference ships (tests/golden/ref_ckpt/, data) reproduces what the real
    reference computed from the same files (tests/golden/make_ckpt_golden.py): trained weights, not only seeded ones."""
    fix = np.load(os.path.join(golden_dir, "ref_ckpt_outputs.npz"))
    d = os.path.join(golden_dir, "ref_ckpt", alg)
    ld = lambda k: {n: t.numpy() for n, t in torch.load(os.path.join(d, k + "_params.pkl"), map_location="cpu").items()}
    B, T = 3, 5
    args = seeded.make_args("2s3z", alg, episode_limit=T)
    agent = ld("rnn_net")
    mixer = ld("mixer_net") if alg != "vdn" else {}
    v = ld("v_net") if alg == "qtran_base" else None
    extra = seeded.seeded_state(seeded.qmix_param_shapes(args), seed=14) if alg == "qtran_base" else None
    st = learners.LearnerState(args, agent, mixer, v, extra)
    batch = seeded.make_batch(args, B, seed=700, lengths=[5, 3, -1])
    assert abs(seeded.checksum(batch) - float(fix[alg + "/batch_checksum"])) < 1e-6
    bt = learners.to_tensors(batch, T)
    tol = dict(atol=2e-5, rtol=1e-5)
    with torch.no_grad():
        h0 = torch.zeros(B * args.n_agents, args.rnn_hidden_dim)
        q, h, _ = nets.agent_unroll(st.agent, bt["o"], nets.shifted_onehot(bt["u_onehot"]), h0)
        np.testing.assert_allclose(q.numpy(), fix[alg + "/q_cur"], **tol)
        np.testing.assert_allclose(h.numpy(), fix[alg + "/h_cur"], **tol)
        qc = torch.gather(q, 3, bt["u"]).squeeze(3)
        if alg == "vdn":
            np.testing.assert_allclose(nets.vdn(qc).numpy(), fix[alg + "/q_tot"], **tol)
        elif alg == "qplex":
            qd = q.clone(); qd[bt["avail_u"] == 0] = learners.MASK_BIG
            np.testing.assert_allclose(nets.qplex(st.mixer, qc, bt["s"], args, is_v=True).numpy(), fix[alg + "/v_tot"], **tol)
            np.testing.assert_allclose(nets.qplex(st.mixer, qc, bt["s"], args, actions=bt["u_onehot"],
                                                  max_q_i=qd.max(dim=3)[0]).numpy(), fix[alg + "/a_tot"], **tol)
        else:
            np.testing.assert_allclose(nets.qtran_q(st.mixer, bt["s"], h, bt["u_onehot"], args).numpy(), fix[alg + "/joint_q"], **tol)
            np.testing.assert_allclose(nets.qtran_v(st.v, bt["s"], h, args).numpy(), fix[alg + "/v"], **tol)
