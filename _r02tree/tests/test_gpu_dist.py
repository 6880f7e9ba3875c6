"""Data-parallel path on the GPU: two / three ranks (all on GPU 0, gloo backend moving CUDA tensors) must
reproduce the single-process update on the same global batch - loss, max_episode_len and the
parameters after 3 updates - and bench.py must run under torch.distributed.run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import seeded, learners

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("alg,shape,world", [("qmix", "2s3z", 2), ("qtran_base", "3s5z", 2), ("qplex", "2s3z", 3)])
def test_ranks_equal_one_process(alg, shape, world):
    from test_gpu_learners import build_product
    port = 29500 + (os.getpid() % 400)
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                "--master-addr", "127.0.0.1", "--master-port", str(port), "tests/dist_parity_worker.py", alg, shape],
               {"MARL_BENCH_BACKEND": "gloo"})
    B, T = 6, 6
    lengths = [6, 2, 3, 4, 2, 3]
    case = ("x", shape, alg, B, T, lengths, {})
    args, mac, learner = build_product(case)
    losses = []
    for i in range(3):
        losses.append(learner.train(seeded.make_batch(args, B, seed=100 + i, lengths=lengths), i))
    np.testing.assert_allclose(res["losses"], losses, rtol=2e-5)
    assert res["T"] == learner.max_episode_len
    flat = learner._flat.flat.double().cpu().numpy()
    np.testing.assert_allclose(res["param_sum"], flat.sum(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(res["param_abs"], np.abs(flat).sum(), rtol=1e-6)


def test_bench_runs_under_torchrun_two_ranks():
    port = 29950 + (os.getpid() % 40)
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--envs", "64",
              "--T", "10", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--leg-iters", "1"],
             {"MARL_BENCH_BACKEND": "gloo", "MARL_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 32 and d["value"] > 0
    # (at full size the double-Q unroll reuses the eval unroll's input-side work and is listed beside the other two)
    assert d["scaling"] == "strong" and d["roofline"]["launches_timed"] == 6
    assert d["roofline"]["by_launch"]["reuse"]["launches_timed"] == 2     # the double-Q unroll of each update (pipelined kernel here)
