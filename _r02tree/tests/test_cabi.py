"""The C-ABI library loads on a CPU-only host and exports every symbol include/marl_hip.h declares;
the ctypes table lists exactly those symbols.  No compute is called."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "marl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(marl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from marl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    _lib.load()
    assert b"gfx950" in _lib.load().marl_hip_version()


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle import seeded
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    args = seeded.make_args("2s3z", "qmix", episode_limit=4)
    mac = SharedMAC(args)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        QLearner(mac, args)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mac.init_hidden(2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SyntheticSMACEnv(4, 5, 80, 120, 11, 4)


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "marl_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
