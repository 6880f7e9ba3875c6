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
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_refuses_more_gpus_than_visible(monkeypatch):
    import bench
    monkeypatch.delenv("MARL_BENCH_ONE_DEVICE", raising=False)
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
