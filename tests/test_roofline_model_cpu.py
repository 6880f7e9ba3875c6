"""bench.py's model of what the split (bf16x6) kernels put on the matrix cores - wave-level MFMA instructions per launch, derived from
each kernel's decomposition - against the hardware's own count (SQ_INSTS_MFMA of the committed rocprofv3 PMC passes of the headline
workload): the `roofline` rows of the bench line rest on these counts (within 2 %).  No GPU needed: the PMC file
is data, the launch geometry comes from the library's host-side plan functions."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, T, N, O, S, A, EMB = 4096, 120, 5, 80, 120, 11, 32          # the headline: QMIX, 2s3z shape


def _pmc():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_qmix_2s3z_T120_envs4096_bf16x6.json")))
    if not files:
        pytest.skip("no PMC file of the headline workload under profiles/")
    return json.load(open(files[-1]))["kernels"], os.path.basename(files[-1])


def _close(model, counted, what):
    assert counted > 0 and abs(model - counted) <= 0.02 * counted, "%s: model %d instructions, PMC %d" % (what, model, counted)


def test_bptt_model_matches_the_counters():
    import bench
    k, f = _pmc()
    two = [v for n, v in k.items() if re.match(r"agent_bwd_x6_kernel<.*, 2>", n)]
    one = [v for n, v in k.items() if re.match(r"agent_bwd_x6_kernel<.*, 1>", n)]
    n2, n1 = bench.bptt_x6_plan(E * N)
    assert (n2, n1) == (512, 256) and len(two) == 1 and len(one) == 1
    m = bench.mfma_bptt_x6(E, T, N)
    _close(T * n2 * 1176, two[0]["SQ_INSTS_MFMA"], f + ": two-tile workgroups")
    _close(T * n1 * 888, one[0]["SQ_INSTS_MFMA"], f + ": one-tile workgroups")
    _close(m["k32"] + m["k16"], two[0]["SQ_INSTS_MFMA"] + one[0]["SQ_INSTS_MFMA"], f + ": BPTT")
    # the fp32-equivalent work of those instructions and the useful part the bench row quotes as executed_flop
    H = 64
    useful = (8 * 3 * H * H + 2 * A * H + 2 * H) * E * T * N
    assert 0.97 < useful / (bench.mfma_flop(m) / 6.0) <= 1.0       # (the action dimension of dW_2 is padded from 11 to 16)


def test_rollout_model_matches_the_counters():
    import bench
    from marl_amd import ops, _lib
    k, f = _pmc()
    hit = [(n, v) for n, v in k.items() if n.startswith("synth_rollout_x6_kernel<")]
    assert len(hit) == 1
    rtc, nk1 = [int(x) for x in re.search(r"<(\d+), (\d+)", hit[0][0]).groups()]      # <row tiles, fc1 chunks[, action tiles]>
    lib = _lib.load()
    try:
        plans = []
        for sw in (0, 1, 2):            # the library's own choice, then each decomposition forced: the file says which one it profiled
            assert lib.marl_experiment_set(b"rollout_v1", sw) == 0
            plans.append(ops.synth_rollout_x6_plan(E, N, O, A))
    finally:
        lib.marl_experiment_set(b"rollout_v1", 0)
    plan = [p for p in plans if p[2] == rtc and p[4] == nk1]
    assert plan, (plans, rtc, nk1)
    m = bench.mfma_rollout_x6(plan[0], T)
    _close(m["k32"], hit[0][1]["SQ_INSTS_MFMA"], f + ": whole rollout " + hit[0][0])


def test_qmix_and_unroll_models_match_the_counters():
    import bench
    k, f = _pmc()
    rows = E * T
    fwd = [v for n, v in k.items() if n.startswith("qmix_fused_kernel<false")]
    bwd = [v for n, v in k.items() if n.startswith("qmix_fused_kernel<true")]
    assert len(fwd) == 1 and len(bwd) == 1
    _close(sum(bench.mfma_qmix_x6(rows, N, S, EMB, False).values()), fwd[0]["SQ_INSTS_MFMA"], f + ": QMIX forward")
    _close(sum(bench.mfma_qmix_x6(rows, N, S, EMB, True).values()), bwd[0]["SQ_INSTS_MFMA"], f + ": QMIX loss + backward")
    full = [v for n, v in k.items() if re.match(r"agent_fwd_x6_kernel<\d, (true|false), false,", n)]       # XS = false: plain and saving unrolls
    r6 = [v for n, v in k.items() if n.startswith("agent_fwd_x6p_kernel<")]       # round 6: the plain (target) unroll of large batches
    assert len(full) + len(r6) == 2 and len(r6) <= 1
    for v in full:
        _close(bench.mfma_unroll_x6(E, T, N)["k32"], v["SQ_INSTS_MFMA"], f + ": unroll")
    for v in r6:
        _close(bench.mfma_unroll_x6(E, T, N, r6=True)["k32"], v["SQ_INSTS_MFMA"], f + ": plain unroll, round-6 decomposition")
