"""One small invocation of the hot path on cuda:0 checked against the CPU oracle
(called by __graft_entry__.smoke())."""
import numpy as np
import torch


def run():
    from oracle import seeded, learners, rollout as orl
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    T, E = 6, 16
    args = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=0.3, seed=1)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11, scale=2.0)
    mixer = seeded.seeded_state(seeded.qmix_param_shapes(args), seed=12)
    t = lambda d: {k: torch.tensor(v) for k, v in d.items()}
    mac = SharedMAC(args)
    mac.agent.load_state_dict(t(agent))
    learner = QLearner(mac, args)
    learner.mixer.load_state_dict(t(mixer))
    learner.target_mixer.load_state_dict(t(mixer))
    env = SyntheticSMACEnv(E, 5, 80, 120, 11, T, seed=3)
    ep, rew, wins, steps = RolloutWorker(env, mac, args).generate_episodes(E)
    sy = orl.SynthSMAC(5, 80, 120, 11, T, seed=3)
    oep, _, _, osteps, _ = orl.batched_rollout(agent, args, sy, E, 0.3, rseed=1)
    d = ep.numpy()
    assert steps == osteps and np.array_equal(d["u"], oep["u"]), "rollout mismatch vs oracle"
    loss = learner.train(ep, 0)
    st = learners.LearnerState(args, agent, mixer)
    oloss, _, _ = learners.train(st, d, 0)
    assert abs(loss - oloss) <= 1e-4 * max(1.0, abs(oloss)), (loss, oloss)
    # the same update with the agent unrolls and BPTT on the split kernels (opt-in args.gemm_mode = "bf16x6"): same bound
    args6 = seeded.make_args("2s3z", "qmix", episode_limit=T, epsilon=0.3, seed=1)
    args6.gemm_mode = "bf16x6"
    mac6 = SharedMAC(args6)
    mac6.agent.load_state_dict(t(agent))
    learner6 = QLearner(mac6, args6)
    learner6.mixer.load_state_dict(t(mixer))
    learner6.target_mixer.load_state_dict(t(mixer))
    loss6 = learner6.train(ep, 0)
    assert abs(loss6 - oloss) <= 1e-4 * max(1.0, abs(oloss)), (loss6, oloss)
    print("smoke ok: steps=%d loss=%.6f (bf16x6 mode: %.6f) oracle=%.6f" % (steps, loss, loss6, oloss))
