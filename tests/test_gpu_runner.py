"""Runner / launcher / matrix-game harness on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_runner_loop_synthetic(tmp_path):
    from marl_amd.main import build
    from marl_amd.runner import Runner
    from marl_amd.utils.logging import Logger
    args, env = build(["--alg", "qmix", "--map", "2s3z", "--n_envs", "16", "--n_steps", "6000",
                       "--result_dir", str(tmp_path / "res"), "--model_dir", str(tmp_path / "model"),
                       "--evaluate_epoch", "16", "--evaluate_cycle", "3000"])
    args.save_cycle = 2
    log = Logger()
    runner = Runner(env, log, args)
    loss = runner.run(0)
    assert np.isfinite(loss)
    assert len(log.stats["total_loss"]) >= 3 and len(runner.eval_episode_rewards) >= 3
    assert (tmp_path / "model" / "qmix" / "2s3z" / "1_rnn_net_params.pkl").exists()
    steps = [t for t, _ in log.stats["episode_length"]]
    assert steps == sorted(steps) and steps[-1] >= 6000
    # epsilon annealed once per lock-step
    assert runner.rolloutWorker.epsilon < 1.0


@pytest.mark.parametrize("alg,iters", [("qtran_base", 3000), ("qplex", 3000)])
def test_matrix_game_finds_the_optimal_joint_action(alg, iters):
    """payoff [[8,-12,-12],[-12,0,0],[-12,0,0]]: QPLEX / QTRAN-base reach joint action [0,0] (reward 8) as
    in the reference's result/*/MatrixGame/episode_rewards.npy."""
    from marl_amd.matrix_game_test import run
    torch.manual_seed(0)
    q_tot, joint, individual, loss = run(alg, iters, verbose=False)
    assert joint == [0, 0], (q_tot, joint)
    # the learned value is still approaching 8 after 3000 updates; how close it gets by then depends on fp32 summation
    # order (SGD is chaotic), the greedy joint action does not
    assert abs(q_tot[0, 0] - 8.0) < 2.5
