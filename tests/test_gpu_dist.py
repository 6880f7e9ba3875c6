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
    for k in [k for k, v in env.items() if v is None]:      # None = must be unset in the child
        del env[k]
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _ranks_equal_one_process(alg, shape, world, backend, gemm_mode="f32"):
    from test_gpu_learners import build_product
    port = 29500 + (os.getpid() % 400)
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                "--master-addr", "127.0.0.1", "--master-port", str(port), "tests/dist_parity_worker.py", alg, shape, gemm_mode],
               {"MARL_BENCH_BACKEND": backend})
    B, T = 6, 6
    lengths = [6, 2, 3, 4, 2, 3]
    case = ("x", shape, alg, B, T, lengths, {})
    args, mac, learner = build_product(case, gemm_mode)
    losses = []
    for i in range(3):
        losses.append(learner.train(seeded.make_batch(args, B, seed=100 + i, lengths=lengths), i))
    np.testing.assert_allclose(res["losses"], losses, rtol=2e-5)
    assert res["T"] == learner.max_episode_len
    flat = learner._flat.flat.double().cpu().numpy()
    np.testing.assert_allclose(res["param_sum"], flat.sum(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(res["param_abs"], np.abs(flat).sum(), rtol=1e-6)


@pytest.mark.parametrize("alg,shape,world,gemm_mode", [("qmix", "2s3z", 2, "f32"), ("qtran_base", "3s5z", 2, "f32"), ("qplex", "2s3z", 3, "f32"),
                                                       ("qmix", "2s3z", 2, "bf16x6"), ("qplex", "2s3z", 3, "bf16x6")])
def test_ranks_equal_one_process(alg, shape, world, gemm_mode):
    _ranks_equal_one_process(alg, shape, world, "gloo", gemm_mode)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with two ranks needs two GPUs (the 1-GPU box runs the gloo variant)")
@pytest.mark.parametrize("alg,shape", [("qmix", "2s3z"), ("qtran_base", "3s5z")])
def test_ranks_equal_one_process_rccl(alg, shape):
    """the same comparison with one rank per GPU over RCCL (backend "nccl"): gradient all-reduce, MAX all-reduce of
    max_episode_len and the rank-0 broadcast cross xGMI"""
    _ranks_equal_one_process(alg, shape, 2, "nccl")


def test_bench_runs_under_torchrun_two_ranks():
    port = 29950 + (os.getpid() % 40)
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--envs", "64",
              "--T", "10", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--leg-iters", "1", "--gemm-mode", "f32", "--hip-graph", "off"],
             {"MARL_BENCH_BACKEND": "gloo", "MARL_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 32 and d["value"] > 0 and d["dtype"] == "f32" and "f32_mfma_twin" not in d
    # the roofline object describes the kernel with the largest total time; every timed kernel is listed with its launches
    assert d["scaling"] == "strong" and d["roofline"]["frac"] > 0 and d["roofline"]["launches_timed"] >= 2
    names = {k["name"].split("[")[0].split(" ")[0]: k["launches_timed"] for k in d["roofline"]["kernels"]}
    assert names.get("agent_fwd_kernel") == 2 and names.get("agent_bwd_kernel") == 2, names      # (one entry per unroll kind, 2 steps)
    assert d["rccl"] == {"backend": "gloo", "world_seen": 2}


def test_bench_launches_its_own_ranks():
    """plain `python bench.py --gpus 2` (no torchrun, the form the driver uses): bench.py starts the two ranks itself as a
    fresh child before it touches the GPU, and rank 0's line comes out on its stdout"""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--envs", "64", "--T", "10", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--leg-iters", "1"],
             {"MARL_BENCH_BACKEND": "gloo", "MARL_BENCH_ONE_DEVICE": "1", "WORLD_SIZE": None})
    assert d["n_gpus"] == 2 and d["rccl"]["world_seen"] == 2 and d["config"]["envs_per_gpu"] == 32 and d["value"] > 0
    # the default arithmetic is the split mode, with the fp32-MFMA twin of the same timed region beside it
    assert d["config"]["gemm_mode"] == "bf16x6" and d["dtype"].startswith("f32 (bf16x6")
    assert any("x6" in k["rocprof_name"] for k in d["roofline"]["kernels"]), d["roofline"]["kernels"]
    assert d["f32_mfma_twin"]["value"] > 0 and d["f32_mfma_twin"]["dtype"] == "f32" and d["f32_mfma_twin"]["roofline"]["frac"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_launches_its_own_ranks_rccl():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--envs", "64", "--T", "10", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--leg-iters", "1"], {"WORLD_SIZE": None})
    assert d["n_gpus"] == 2 and d["rccl"] == {"backend": "nccl", "world_seen": 2}
