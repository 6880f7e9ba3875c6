"""Process-level switches live in ONE place (marl_amd/experiments.py), are read from the environment once, and reach the library
through marl_experiment_set - no launch path calls getenv (host functions only: no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_switch_table_roundtrip():
    from marl_amd import _lib, experiments
    lib = _lib.load()
    for name, default in experiments.LIB_DEFAULTS.items():
        assert lib.marl_experiment_get(name.encode()) == experiments.get(name)
    assert lib.marl_experiment_set(b"no_such_switch", 1) == -1 and lib.marl_experiment_get(b"no_such_switch") == -2 ** 31
    with experiments.override(fwd_dma=1, bwd_pipe_max_rt=2, no_chain=1):
        assert lib.marl_experiment_get(b"fwd_dma") == 1 and lib.marl_experiment_get(b"bwd_pipe_max_rt") == 2
        assert experiments.get("no_chain") == 1
        # the wide-state forward's kernel choice follows the table (a host function of the C-ABI)
        experiments.set("wide_res", 0)
        assert lib.marl_qmix_wide_fwd_kernel(122880, 10, 322, 1).decode().startswith("qmix_wide_kernel<false")
        experiments.set("wide_res", 1)
        assert lib.marl_qmix_wide_fwd_kernel(122880, 10, 322, 1).decode() == "qmix_wide_res_fwd_kernel"
    assert lib.marl_experiment_get(b"fwd_dma") == 0 and lib.marl_experiment_get(b"bwd_pipe_max_rt") == 4 and experiments.get("no_chain") == 0


def test_environment_is_read_once_at_import():
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['MARL_FWD_XS'] = '0'; os.environ['MARL_CHAIN_SPLIT'] = '160';"
            "from marl_amd import experiments, _lib; lib = _lib.load();"
            "os.environ['MARL_FWD_XS'] = '1'; os.environ['MARL_NO_PAIR'] = '1';"      # too late: not re-read
            "print(experiments.get('fwd_xs'), lib.marl_experiment_get(b'fwd_xs'), lib.marl_agent_unroll_reuse_supported(512, 120, 5, 80, 11, 0),"
            " experiments.get('chain_split'), experiments.get('no_pair'))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["0", "0", "0", "160", "0"], out.stdout


def test_no_getenv_in_kernel_sources_and_one_reader_in_the_host_package():
    csrc = os.path.join(ROOT, "marl_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    readers = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "marl_amd")):
        for f in fs:
            if f.endswith(".py") and re.search(r"os\.environ\.get\(\s*[\"']MARL_(?!HIP_LIB|N_ENVS|ENV_THREADS)", open(os.path.join(dp, f)).read()):
                readers.append(os.path.relpath(os.path.join(dp, f), ROOT))
    assert readers == ["marl_amd/experiments.py"], readers           # (MARL_HIP_LIB / MARL_N_ENVS / MARL_ENV_THREADS are not switches)


def test_graph_policy_from_args_and_split_step_model():
    """GraphedUpdate.from_args: absent / None / "auto" -> auto (small batches), True -> always, False -> never; PairedUnroll's
    step-time model per arithmetic (host logic only)."""
    import types
    from marl_amd.algorithm.common import GraphedUpdate, PairedUnroll
    g = GraphedUpdate.from_args(types.SimpleNamespace())
    assert g is not None and g.auto and GraphedUpdate.AUTO_MAX_EPISODES == 1536
    assert GraphedUpdate.from_args(types.SimpleNamespace(hip_graph="auto")).auto
    assert GraphedUpdate.from_args(types.SimpleNamespace(hip_graph=True)).auto is False
    assert GraphedUpdate.from_args(types.SimpleNamespace(hip_graph=False)) is None
    f32, x6 = PairedUnroll(), PairedUnroll(x6=True)
    assert f32.chain_split(512 * 5, 120, 80) == (160, 96)              # the fp32 kernels' chain schedule at the 512-env shard
    assert x6._step_us(1) < f32._step_us(1) and x6._step_us(5) < f32._step_us(5)
    assert x6.chain_split(4096 * 5, 120, 80) is None and f32.chain_split(4096 * 5, 120, 80) is None      # large batches: plain schedule


def test_get_answers_library_switches_from_the_library_and_set_bumps_the_generation():
    """a C-ABI caller may flip a library switch through marl_experiment_set directly - the host's decisions read the
    library's table, not a Python mirror; every experiments.set() moves the generation a captured hipGraph schedule is keyed by."""
    from marl_amd import experiments, _lib
    lib = _lib.load()
    try:
        assert experiments.get("fwd_xs") == 1 and experiments.get("rollout_v1") == 0 and experiments.get("unroll_r6") == 1
        assert lib.marl_experiment_set(b"fwd_xs", 0) == 0
        assert experiments.get("fwd_xs") == 0                     # not the value this module last wrote
        g0 = experiments.generation
        with experiments.override(rollout_v1=2, big_pair=1):
            assert lib.marl_experiment_get(b"rollout_v1") == 2 and experiments.get("big_pair") == 1 and experiments.generation > g0
        assert lib.marl_experiment_get(b"rollout_v1") == 0 and experiments.get("big_pair") == 0
        assert lib.marl_experiment_set(b"no_such_switch", 1) != 0
    finally:
        lib.marl_experiment_set(b"fwd_xs", 1)
