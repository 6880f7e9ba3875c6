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


def _src_cpu(k):
    """a marl_src_t describing a dense (rows, k) input with 16-byte aligned rows (widths only: no device memory is touched)"""
    from marl_amd._lib import MarlSrc
    s = MarlSrc()
    s.p0, s.ld0, s.k0 = 4096, (k + 3) // 4 * 4, k
    return s


def test_split_kernel_capability_predicates():
    """The *_supported entry points of the bf16x6 kernels are host functions (no GPU): which agent shapes the split unroll and the
    split BPTT cover - the three map sizes of BASELINE's configurations - and what they leave to the fp32 kernels."""
    from marl_amd import _lib
    lib = _lib.load()
    fwd = lambda B, T, N, O, A, la=1, ru=1: lib.marl_agent_unroll_x6_supported(B, T, N, O, A, la, ru)
    bwd = lambda B, T, N, A, sparse=1: lib.marl_agent_unroll_bwd_x6_supported(B, T, N, A, sparse)
    assert fwd(512, 120, 5, 80, 11) == 1            # 2s3z: 96 input columns, three fc1 chunks
    assert fwd(512, 150, 8, 128, 14) == 1           # 3s5z: 150 columns, five chunks
    assert fwd(1024, 120, 10, 176, 18) == 1         # MMM2: 204 columns, seven chunks, two action tiles
    assert fwd(512, 3, 5, 80, 11) == 0              # T < 4: no pipeline to fill - the fp32 kernels
    assert fwd(512, 120, 5, 80, 18) == 0            # more than 16 actions on a narrow input: one action tile only
    assert fwd(512, 120, 10, 240, 18) == 0          # wider than 224 input columns
    assert fwd(512, 120, 5, 84, 11) == 0            # observation width not a multiple of 8
    assert fwd(512, 120, 5, 192, 11) == 1 and fwd(512, 120, 5, 200, 11) == 0      # O <= 192: three prefetch registers per thread of one team
    assert fwd(512, 120, 8, 128, 16) == 1 and fwd(512, 120, 8, 128, 17) == 0      # 152 / 153 input columns: one action tile up to 160 columns
    assert fwd(512, 120, 10, 176, 32) == 1 and fwd(512, 120, 10, 176, 33) == 0    # wider than 160: two action tiles, 32 actions at most
    # the split whole-rollout kernel: 2s3z-, 3s5z- and MMM2-sized agents (<= 224 input columns; <= 16 actions up to 160 columns, <= 32
    # beyond: two action tiles; whole environments in at most five - wide inputs: four / three - row tiles of 16 rows, an environment's
    # agents on one wave)
    rx6 = lib.marl_synth_rollout_x6_supported
    assert rx6(5, 80, 11) == 1 and rx6(8, 128, 14) == 1 and rx6(2, 4, 3) == 1 and rx6(49, 40, 5) == 1 and rx6(33, 100, 5) == 1
    assert rx6(10, 176, 18) == 1 and rx6(4, 180, 32) == 1 and rx6(4, 180, 33) == 0 and rx6(10, 200, 18) == 0
    assert rx6(5, 80, 17) == 0 and rx6(5, 82, 11) == 0 and rx6(65, 40, 5) == 0 and rx6(64, 40, 5) == 1 and rx6(49, 176, 18) == 0
    # ... and how a batch runs: the round-5 decomposition while it holds one row tile per workgroup, the round-6 one (up to five tiles)
    # beyond - (decomposition, workgroups, row tiles, environments per workgroup, fc1 chunks)
    import ctypes
    def plan(E, N, O, A):
        out = (ctypes.c_int * 5)()
        assert lib.marl_synth_rollout_x6_plan(E, N, O, A, 1, 1, out) == 0
        return tuple(out)
    assert plan(512, 5, 80, 11) == (1, 256, 1, 2, 3) and plan(768, 5, 80, 11) == (1, 256, 1, 3, 3) and plan(1024, 5, 80, 11) == (2, 256, 2, 4, 3)
    assert plan(2304, 5, 80, 11) == (2, 256, 3, 9, 3)
    assert plan(4096, 5, 80, 11) == (2, 256, 5, 16, 3) and plan(8192, 5, 80, 11) == (2, 512, 5, 16, 3)
    assert plan(2048, 8, 128, 14) == (2, 256, 4, 8, 5) and plan(100, 49, 40, 5) == (2, 100, 4, 1, 3)
    assert plan(1024, 10, 176, 18) == (2, 256, 3, 4, 7) and plan(2048, 10, 176, 18) == (2, 512, 3, 4, 7)      # MMM2: three tiles at most
    assert lib.marl_synth_rollout_x6_plan(512, 10, 200, 18, 1, 1, (ctypes.c_int * 5)()) != 0
    # the fused-head split pair: the padded input width must leave a free column in its last 64-column block
    m3 = lambda k, n3=1: lib.marl_mlp3_x6_supported(__import__("ctypes").byref(_src_cpu(k)), k, 64, 64, n3, 10)
    assert m3(112) == 1 and m3(120) == 1 and m3(175) == 1 and m3(188) == 1
    assert m3(191) == 0                             # a dense input of 191 columns is padded to 192 by the kernels: no free column
    assert m3(128) == 0 and m3(192) == 0 and m3(64) == 0 and m3(120, 17) == 0
    assert bwd(512, 120, 5, 11) == 1 and bwd(1024, 120, 10, 18) == 1 and bwd(512, 150, 8, 14) == 1
    assert bwd(512, 120, 5, 11, 0) == 0             # a dense dq: the fp32 kernels
    assert bwd(512, 2, 5, 11) == 0 and bwd(512, 120, 5, 40) == 0
    # workspace: one slab per workgroup - one row tile (16 rows) per workgroup up to 256 tiles; beyond that two-tile workgroups in
    # full rounds of 256, and what is left as one-tile workgroups when that is at most one round, else two-tile workgroups throughout
    slab = 4 * (2 * 192 * 64 + 11 * 64 + 2 * 192 + 11)
    assert lib.marl_agent_bwd_x6_workspace(512, 5, 11) == (512 * 5 // 16) * slab
    assert lib.marl_agent_bwd_x6_workspace(4096, 5, 11) == (512 + 256) * slab          # 1280 tiles = 2 x 512 + 256
    assert lib.marl_agent_bwd_x6_workspace(2048, 5, 11) == (256 + 128) * slab          # 640 tiles = 512 + 128
    assert lib.marl_agent_bwd_x6_workspace(3000, 5, 11) == 469 * slab                  # 938 tiles = 512 + 426: two-tile workgroups throughout
    assert lib.marl_agent_bwd_x6_workspace(1024, 5, 11) == 160 * slab                  # 320 tiles: less than one full round of two-tile workgroups


def test_qtran_row_kernel_predicates_and_workspace():
    """Host functions of the QTRAN row-level kernels (no GPU): which state widths / encoder widths they cover, and the slab
    workspace the row-gradient kernel asks for (256 workgroups x [64 x 16 ceil(S/16) | 64 x AEP | 64 x 64 | AEP x AEP | 256])."""
    from marl_amd import _lib
    lib = _lib.load()
    sp, wg, ws = lib.marl_qtran_state_parts_supported, lib.marl_qtran_wgrad_rows_supported, lib.marl_qtran_wgrad_rows_workspace
    assert sp(216) and sp(120) and sp(322) and sp(1) and sp(384)          # (S % 4 != 0: padded rows, checked per call)
    assert not sp(0) and not sp(385)
    assert wg(216, 78) and wg(120, 64) and wg(384, 80) and wg(4, 64)
    assert not wg(322, 74) and not wg(388, 78) and not wg(216, 81) and not wg(216, 48)
    for S, AE in ((216, 78), (120, 64), (384, 80)):
        aep, ts = (AE + 15) // 16 * 16, (S + 15) // 16
        assert ws(S, AE) == 256 * (64 * 16 * ts + 64 * aep + 64 * 64 + aep * aep + 256) * 4
