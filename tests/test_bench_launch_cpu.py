"""bench.py --gpus N without a launcher starts torch.distributed.run as a child (no GPU needed to check the command)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command(monkeypatch):
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("MARL_BENCH_ONE_DEVICE", "1")        # (no GPUs here: skip the device-count check)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.self_launch(4) == 7                          # the child's return code is ours
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    # rendezvous on 127.0.0.1 at a port torchrun picks itself (no bind-then-close race with other jobs of the node)
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_honours_a_given_master_port(monkeypatch):
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: seen.setdefault("cmd", cmd) and subprocess.CompletedProcess(cmd, 0))
    monkeypatch.setenv("MARL_BENCH_ONE_DEVICE", "1")
    monkeypatch.setenv("MASTER_PORT", "29777")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert bench.self_launch(2) == 0
    cmd = seen["cmd"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777" and "--standalone" not in cmd


def test_visible_gpus_never_touches_the_runtime(monkeypatch):
    """the parent of a self-launch counts GPUs from the environment / the KFD topology, not through torch.cuda"""
    import bench
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("must not ask the runtime")))
    for k in ("ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):      # (a GPU box's lease sets one of them)
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")          # HIP_ indexes into the ROCR_ set: both lists bound the count
    assert bench.visible_gpus() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    for k in ("ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert bench.visible_gpus() in (None, 0) or bench.visible_gpus() > 0        # (no KFD in the build container: None)


def test_self_launch_refuses_more_gpus_than_visible(monkeypatch):
    import bench
    monkeypatch.delenv("MARL_BENCH_ONE_DEVICE", raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not launch")))
    assert bench.self_launch(64) == 2


def test_plain_invocation_with_gpus_gt_1_takes_the_launch_path(monkeypatch):
    import bench
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda n: called.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == [8]


def test_pmc_traffic_picks_the_instantiation_a_timed_call_launched():
    """bench.py matches a timed C-ABI call to ONE kernel of the committed PMC file: the three forward unrolls share a name prefix in
    either arithmetic (fp32: by what the launch writes / multiplies; split kernels: by template arguments), the fused heads differ by
    their padded input width, and a name that matches several kernels yields no traffic rather than a wrong one."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    k = lambda b, **kw: dict(hbm_bytes_per_launch=b, **kw)
    pmc = {"agent_fwd_kernel<1, true, true, 4, false, false, false, false>": k(6.7e9, WRITE_SIZE=5e6, SQ_INSTS_MFMA=9e6),
           "agent_fwd_kernel<1, false, true, 4, false, false, false, false>": k(2.0e9, WRITE_SIZE=1e5, SQ_INSTS_MFMA=9e6),
           "agent_fwd_kernel<1, false, true, 4, true, false, false, false>": k(2.1e9, WRITE_SIZE=1e5, SQ_INSTS_MFMA=4e6),
           "agent_fwd_x6_kernel<2, true, false, true, 7, 2>": k(6.8e9), "agent_fwd_x6_kernel<2, false, false, false, 7, 2>": k(2.0e9),
           "agent_fwd_x6_kernel<2, false, true, false, 7, 2>": k(2.03e9),
           "agent_bwd_kernel<1, false, 1>": k(5.2e9), "agent_bwd_x6_kernel<false, false, 2>": k(4.5e9),
           "mlp3x6_fwd_kernel<8, true, 6>": k(1.5e9), "mlp3x6_fwd_kernel<12, true, 6>": k(1.9e9)}
    e = lambda name, roc: {"name": name, "rocprof_name": roc}
    assert bench.pmc_traffic(pmc, e("agent_fwd_kernel[save: eval unroll ...]", "agent_fwd"))["hbm_bytes_per_launch"] == 6.7e9
    assert bench.pmc_traffic(pmc, e("agent_fwd_kernel[plain: target unroll]", "agent_fwd"))["hbm_bytes_per_launch"] == 2.0e9
    assert bench.pmc_traffic(pmc, e("agent_fwd_kernel[reuse: double-Q unroll ...]", "agent_fwd"))["hbm_bytes_per_launch"] == 2.1e9
    assert bench.pmc_traffic(pmc, e("agent_fwd_x6_kernel[save: eval unroll ...]", "agent_fwd_x6"))["hbm_bytes_per_launch"] == 6.8e9
    assert bench.pmc_traffic(pmc, e("agent_fwd_x6_kernel[plain: target unroll]", "agent_fwd_x6"))["hbm_bytes_per_launch"] == 2.0e9
    assert bench.pmc_traffic(pmc, e("agent_fwd_x6_kernel[reuse: double-Q unroll ...]", "agent_fwd_x6"))["hbm_bytes_per_launch"] == 2.03e9
    assert bench.pmc_traffic(pmc, e("agent_bwd_kernel (BPTT ...)", "agent_bwd_kernel"))["hbm_bytes_per_launch"] == 5.2e9
    assert bench.pmc_traffic(pmc, e("agent_bwd_x6_kernel (BPTT ...)", "agent_bwd_x6_kernel"))["hbm_bytes_per_launch"] == 4.5e9
    assert bench.pmc_traffic(pmc, e("mlp3x6_fwd_kernel (fused 64-wide heads, 10 heads, K1=175)", "mlp3x6_fwd"))["hbm_bytes_per_launch"] == 1.9e9
    assert bench.pmc_traffic(pmc, e("mlp3x6_fwd_kernel (fused 64-wide heads, 10 heads, K1=120)", "mlp3x6_fwd"))["hbm_bytes_per_launch"] == 1.5e9
    assert bench.pmc_traffic({}, e("agent_bwd_kernel (BPTT ...)", "agent_bwd_kernel")) is None
    # the split BPTT call is two launches at large batches (two-tile workgroups in full rounds + a round of one-tile ones): summed
    pmc2 = dict(pmc)
    pmc2["agent_bwd_x6_kernel<false, false, 1>"] = k(0.9e9)
    assert bench.pmc_traffic(pmc2, e("agent_bwd_x6_kernel (BPTT ...)", "agent_bwd_x6_kernel"))["hbm_bytes_per_launch"] == 5.4e9
