#!/usr/bin/env python3
"""Headline benchmark: env-steps/s and learner updates/s of the MARL hot path on MI355X.

One "step" = the reference runner's inner iteration (runner.py:85-98) at scale: batched rollout of
the rank's envs for T lock-steps -> ReplayBuffer.store_episode -> sample -> one learner.train().
Workload (BASELINE.json metric): QMIX, synthetic 2s3z shape (N=5, O=80, S=120, A=11, T=120),
4096 envs GLOBAL, split over the ranks (strong scaling); gradients all-reduced over RCCL.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import gc
import json
import os
import sys
import time
import types

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / tensor sharing across ranks)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPES = {"2s3z": (5, 80, 120, 11, 120), "3s5z": (8, 128, 216, 14, 150), "MMM2": (10, 176, 322, 18, 120)}
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak; a bf16x6 split spends six bf16 products per fp32 product


def make_args(alg, shape, T):
    from marl_amd.common.arguments import get_mixer_args
    N, O, S, A, T0 = SHAPES[shape]
    a = types.SimpleNamespace(alg=alg, map=shape, n_agents=N, obs_shape=O, state_shape=S, n_actions=A,
                              episode_limit=T or T0, last_action=True, reuse_network=True, gamma=0.99,
                              optimizer="RMS", cuda=True, RTW=False, load_model=False, model_dir="./model",
                              result_dir="./result", replay_dir="", n_episodes=1, evaluate_epoch=0, seed=123)
    get_mixer_args(a)
    return a


def agent_flops(a):
    I = a.obs_shape + a.n_actions + a.n_agents
    H = a.rnn_hidden_dim
    return 2 * I * H + 12 * H * H + 2 * H * a.n_actions       # SURVEY 8d F_a


def learner_flops_per_transition(a, alg):
    """dense-layer FLOP (2 x MAC) of one learner update per (episode, step) transition."""
    N, S, A, H, E = a.n_agents, a.state_shape, a.n_actions, a.rnn_hidden_dim, a.qmix_hidden_dim
    Fa = agent_flops(a)
    if alg == "vdn":
        return 5 * N * Fa
    if alg == "qmix":
        Fm = 2 * S * N * E + 3 * 2 * S * E + 2 * E + 2 * N * E + 2 * E
        return 5 * N * Fa + 4 * Fm
    if alg == "qplex":
        K, AE = a.num_kernel, a.adv_hypernet_embed
        trans = 2 * 2 * (S * AE + AE * N)
        lam = K * 2 * ((S * AE + AE * AE + AE) + (S * AE + AE * AE + AE * N) + ((S + N * A) * AE + AE * AE + AE * N))
        return 5 * N * Fa + 4 * (2 * trans + lam)     # SURVEY's figure: 5 N F_a + 2 (32 000 + 823 040) + 2 * 855 040 = 4.99 M on 2s3z
    if alg.startswith("qtran"):
        Q = a.qtran_hidden_dim
        q = N * 2 * 2 * (H + A) ** 2 + 2 * ((S + H + A) * Q + Q * Q + Q)
        v = N * 2 * 2 * H * H + 2 * ((S + H) * Q + Q * Q + Q)
        return 4 * N * Fa + 3 * q + 2 * q + 3 * v
    raise ValueError(alg)


# ---- what the split (bf16x6) kernels put on the matrix cores: wave-level MFMA instructions per launch, from each kernel's decomposition
# (checked against SQ_INSTS_MFMA of the committed PMC passes by tests/test_roofline_model_cpu.py).  "k32" = v_mfma_f32_16x16x32_bf16
# (16 384 FLOP), "k16" = v_mfma_f32_16x16x16_bf16 (8 192 FLOP); an fp32 product costs six of either.  Tile padding and redundant
# products ARE counted here (this is what the pipe executes); `executed_flop` of a row counts the useful products only.
MFMA_FLOP = {"k32": 16384.0, "k16": 8192.0}


def mfma_flop(m):
    return sum(MFMA_FLOP[k] * v for k, v in m.items())


def bptt_x6_plan(rows):
    """(two-tile workgroups, one-tile workgroups) of agent_bwd_x6_kernel for a batch of `rows` (episode, agent) rows
    (csrc/agent_bwd_x6.hip: bx6_plan)"""
    tiles = (rows + 15) // 16
    if tiles <= 256:
        return 0, tiles
    n2 = tiles // 512 * 256
    rem = tiles - 2 * n2
    if n2 > 0 and 0 < rem <= 256:
        return n2, rem
    return (tiles + 1) // 2, 0


def mfma_bptt_x6(B, T, N):
    """per step a two-tile workgroup multiplies: team R  carry' = G W_hh and dx = G W_ih (6 k chunks x 2 tiles x 6 terms each, four slice
    waves) = 576, team I  dW_ih, dW_hh (48 output tiles each) and dW_2 (4: the action dimension padded to 16) x 6 terms = 600 - all on the
    32-deep instruction; a one-tile workgroup: 288 of those and its 600 reductions on the 16-deep one.  dq -> dh is not matrix work (the
    loss reaches q through one or two (action, gradient) pairs per row)."""
    n2, n1 = bptt_x6_plan(B * N)
    return {"k32": T * (n2 * 1176 + n1 * 288), "k16": T * n1 * 600}


def mfma_rollout_x6(plan, T):
    """per lock-step and row tile: recurrence 4 slice waves x 72 (x W_ih + h W_hh: 3 gates x (2 + 2) chunks x 6 terms), fc1 4 x 6 NK1,
    fc2 12 products by EVERY slice wave of a team (each makes the choice of its own rows): 288 + 24 NK1 + 48 - the same count in both
    decompositions (ops.synth_rollout_x6_plan: workgroups, row tiles per workgroup incl. padding rows)"""
    _, wgs, rtc, _, nk1 = plan
    return {"k32": T * wgs * rtc * (288 + 24 * nk1 + 48 * (2 if nk1 == 7 else 1)), "k16": 0}      # (seven chunks: two action tiles of fc2)


def mfma_qmix_x6(rows, N, S, E, backward):
    """per 16-row tile: hypernet GEMM [16 x S] x [S x (N E + 3 E)] on the 32-deep instruction (S padded to 32s) x 6 terms; the backward
    recomputes it and adds dW = d(out)^T s on the 16-deep one (contraction over the tile's 16 rows: (N E + 3 E) / 16 x ceil(S / 16) output
    tiles x 6 terms)"""
    tiles = (rows + 15) // 16
    cols = (N * E + 3 * E + 15) // 16
    m = {"k32": tiles * cols * ((S + 31) // 32) * 6, "k16": 0}
    if backward:
        m["k16"] = tiles * cols * ((S + 31) // 32 * 2) * 6
    return m


def mfma_unroll_x6(B, T, N, nk1=3, r6=False):
    """plain / saving unroll: per step and row tile 4 x (36 recurrence + 36 input gates + 6 NK1 fc1) + 48 fc2 (agent_x6.hip: fc2 by one
    wave per tile, 12 products on each of its two k-chunk chains x 2... = 48); the round-6 plain unroll (agent_x6p.hip) 288 + 72 + 12"""
    per = 288 + 24 * nk1 + (12 if r6 else 48)
    return {"k32": T * ((B * N + 15) // 16) * per, "k16": 0}


class KernelTimers:
    """HIP-event timing of the C-ABI calls of the hot path inside the timed region, on the stream each one is launched
    on (events are recorded on torch's CURRENT stream at the call, which is the side stream for the forked launches).
    Every timed call carries the FLOP it EXECUTES (2 x multiply-adds of the products the launch really computes, tile
    padding not counted) and its algorithmic HBM bytes; `rocname` is the prefix of the kernel's name in a rocprofv3
    kernel trace, so the line can be recomputed from profiles/*_kernel_stats.csv."""

    def __init__(self, ops, args, E):
        self.ops, self.on, self.rec, self._orig = ops, False, {}, {}
        a = args
        N, O, S, A, H, Em = a.n_agents, a.obs_shape, a.state_shape, a.n_actions, a.rnn_hidden_dim, a.qmix_hidden_dim
        I = O + A + N
        Fa = agent_flops(a)
        fa_in = 2 * I * H + 6 * H * H                 # fc1 + x W_ih: what a reuse launch loads instead of computing

        def fwd(ar, kw):
            B, T, N_, A_ = ar[12], ar[13], ar[14], ar[16]
            if T <= 1:
                return None
            rows = B * N_ * T
            if kw.get("gi_in") is not None:
                return ("agent_fwd_kernel[reuse: double-Q unroll reading the eval unroll's input-side gate sums]", "agent_fwd",
                        Fa * rows - fa_in * B * N_ * (T - 1), Fa * rows, 4.0 * rows * (3 * H + A_))
            if ar[11] is not None:
                return ("agent_fwd_kernel[save: eval unroll storing 6 activation planes%s]" % (" + gate sums" if kw.get("gi_out") is not None else ""),
                        "agent_fwd", Fa * rows, Fa * rows, 4.0 * rows * (O + A_ + 6 * H + (3 * H if kw.get("gi_out") is not None else 0)))
            return ("agent_fwd_kernel[plain: target unroll]", "agent_fwd", Fa * rows, Fa * rows, 4.0 * rows * (O + A_))

        def fwd_x6(ar, kw):
            m = fwd(ar, kw)          # same arguments as agent_unroll_fwd; the kernel is agent_fwd_x6_kernel (csrc/agent_x6.hip)
            if m is None:
                return None
            r6 = ar[11] is None and kw.get("gi_in") is None and ar[9] is None and \
                ops.agent_unroll_x6_plain_r6(ar[12], ar[13], ar[14], O, A, cu_budget=kw.get("cu_budget", 0))      # (saved, gi_in, hs all absent)
            mf = mfma_unroll_x6(ar[12], ar[13], ar[14], 3 if I <= 96 else 5 if I <= 160 else 7, r6) if kw.get("gi_in") is None else None
            if r6:      # csrc/agent_x6p.hip: the plain unroll in the round-6 decomposition (five row tiles per workgroup, two barriers per step)
                return ("agent_fwd_x6p_kernel[plain: target unroll, round-6 decomposition] fp32 products as six bf16 MFMA products", "agent_fwd_x6p_kernel") + m[2:] + (True, mf)
            return (m[0].replace("agent_fwd_kernel", "agent_fwd_x6_kernel") + " fp32 products as six bf16 MFMA products", "agent_fwd_x6") + m[2:] + (True, mf)

        def bwd(ar, kw):
            B, T, N_, A_ = ar[8], ar[9], ar[10], ar[11]
            rows = B * T * N_
            f = (8 * 3 * H * H + 4 * A_ * H) * rows   # dx, dh_prev, dW_ih, dW_hh (2*192*64 each) + dq->dh and dW_2 (2*A*64 each)
            if kw.get("x6"):      # csrc/agent_bwd_x6.hip (opt-in gemm_mode)
                # useful products: dx, carry', dW_ih, dW_hh (2 * 192 * 64 each) and dW_2 (2 * A * 64) on the matrix cores; dq -> dh is one or
                # two sparse (action, gradient) pairs per row on the vector unit (2 * 64 each)
                pairs = 2 if (len(ar) > 14 and ar[14] is not None) or kw.get("dq_idx2") is not None else 1
                fx = (8 * 3 * H * H + 2 * A_ * H + pairs * 2 * H) * rows
                return ("agent_bwd_x6_kernel (BPTT: delta pass + dW_ih / dW_hh / dW_2, fp32 products as six bf16 MFMA products)", "agent_bwd_x6_kernel",
                        fx, f, 4.0 * rows * (10 * H + H), True, mfma_bptt_x6(B, T, N_))
            return ("agent_bwd_kernel (BPTT: delta pass + dW_ih / dW_hh / dW_2)", "agent_bwd_kernel", f, f, 4.0 * rows * (10 * H + H))

        def wgrad(ar, kw):
            M, Nn, K = ar[4], ar[5], ar[6]
            f = 2.0 * M * Nn * (K + 1)
            tall = Nn == H and M >= 4096 and K >= O
            return ("linear_wgrad M=%d N=%d K=%d%s" % (M, Nn, K, " (fc1 gradient: wgrad_tall_kernel)" if tall else ""),
                    "wgrad_tall_kernel" if tall else "wgrad_", f, f, 4.0 * M * (Nn + K))

        def lin(ar, kw):
            M, Nn, K = ar[4], ar[5], ar[6]
            f = 2.0 * M * Nn * K
            return ("linear M=%d N=%d K=%d (generic GEMM)" % (M, Nn, K), "linear_kernel", f, f, 4.0 * M * (Nn + K))

        def qmix(mult, label, roc, loss=False):
            def m(ar, kw):
                rows, N_, S_, E_ = ar[-4], ar[-3], ar[-2], ar[-1]
                f = 2.0 * rows * S_ * (N_ * E_ + 3 * E_) * mult
                if kw.get("x6"):      # the split variant: template arguments <BWD, 8 waves, LOSS, X6 = true>
                    name = "qmix_fused_kernel<%s, 8, %s, true>" % ("true" if mult > 1 else "false", "true" if loss else "false")
                    return (label + ", fp32 products as six bf16 MFMA products", name, f, f, 4.0 * rows * (S_ + N_ + 1), True,
                            mfma_qmix_x6(rows, N_, S_, E_, mult > 1))
                return (label, roc, f, f, 4.0 * rows * (S_ + N_ + 1))
            return m

        def qmix_kw(mult, label, roc, rows_at):
            def m(ar, kw):
                rows, N_, S_, E_ = ar[rows_at:rows_at + 4]
                f = 2.0 * rows * S_ * (N_ * E_ + 3 * E_) * mult
                return (label, roc, f, f, 4.0 * rows * (S_ + N_ + 1))
            return m

        def wide_fwd(ar, kw):
            # the kernel this call launches depends on the shape and operand type (resident weights for bf16 at MMM2 sizes):
            # ask the library for its name, so that the entry can be found in profiles/*_kernel_stats.csv
            rows, N_, S_, E_ = ar[4:8]
            roc = ops.qmix_wide_fwd_kernel(rows, N_, S_, bf16=bool(kw.get("bf16")))
            f = 2.0 * rows * S_ * (N_ * E_ + 3 * E_)
            extra = " + pack kernel + memset of q_tot in the timed interval" if "res" in roc else " + pack kernel in the timed interval"
            return ("%s forward (hypernet GEMM + mixing%s)" % (roc.split("<")[0], extra), roc, f, f, 4.0 * rows * (S_ + N_ + 1))

        def mlp3(back):
            def m(ar, kw):
                if back:
                    w, x, dY, grads, M, K1, N3, g = ar[:8]
                else:
                    w, x, Y, M, K1, N3, g = ar[:7]
                three = bool(w.w2)
                kept = kw.get("hsave") is not None      # h1 / h2 travel forward -> backward instead of being recomputed
                x6 = bool(kw.get("x6"))                 # bf16x6 split kernels (csrc/mlp3_x6.hip, opt-in gemm_mode)
                h2 = 64 * 64 if three else 0
                ffw = 2.0 * M * g * (K1 * 64 + h2 + 64 * N3)
                f = 2.0 * M * g * ((1 if kept else 2) * K1 * 64 + (2 if kept else 3) * h2 + 2 * 64 * N3) if back else ffw
                kn = ("mlp3x6_" if x6 else "mlp3_") + ("bwd" if back else "fwd")
                return ("%s_kernel (fused 64-wide heads, %d heads, K1=%d%s%s)" % (kn, g, K1, ", hidden activations kept" if kept else "",
                                                                                  ", fp32 products as six bf16 MFMA products" if x6 else ""),
                        kn, f, ffw * (3 if back else 1),
                        4.0 * M * (K1 + g * N3) + (4.0 * M * g * (128 if three else 64) if kept else 0.0))
            return m

        def roll(ar, kw):
            E_, T_, N_ = ar[9], ar[10], ar[11]
            f = float(Fa) * E_ * T_ * N_
            by = 4.0 * E_ * (T_ + 1) * (N_ * O + S + N_ * A)
            if kw.get("x6"):      # csrc/rollout_x6.hip: the agent step as bf16x6 split products
                return ("synth_rollout_x6_kernel (whole rollout, T lock-steps; fp32 products as six bf16 MFMA products)", "synth_rollout_x6_kernel", f, f, by, True,
                        mfma_rollout_x6(ops.synth_rollout_x6_plan(E_, N_, O, A), T_))
            return ("synth_rollout_kernel (whole rollout, T lock-steps)", "synth_rollout_kernel", f, f, by)

        self.models = {"agent_unroll_fwd": fwd, "agent_unroll_fwd_x6": fwd_x6, "agent_unroll_bwd": bwd, "linear_wgrad": wgrad, "linear": lin,
                       "qmix_fused_fwd": qmix(1, "qmix_fused_kernel forward (target mixer)", "qmix_fused_kernel<false"),
                       "qmix_fused_bwd": qmix(2, "qmix_fused_kernel backward", "qmix_fused_kernel<true"),
                       "qmix_fused_loss_bwd": qmix(2, "qmix_fused_kernel forward + TD loss + backward", "qmix_fused_kernel<true", loss=True),
                       "qmix_wide_fwd": wide_fwd,
                       "qmix_wide_bwd": qmix_kw(2, "qmix_wide backward (recompute + d(out)) + weight-gradient GEMM", "qmix_wide", 6),
                       "qmix_wide_loss_bwd": qmix_kw(2, "qmix_wide forward + TD loss + backward + weight-gradient GEMM", "qmix_wide", 12),
                       "mlp3_fwd": mlp3(False), "mlp3_bwd": mlp3(True), "synth_rollout": roll}
        for name, model in self.models.items():
            self._wrap(name, model)

    def close(self):
        """restore the unwrapped entry points"""
        for name, orig in self._orig.items():
            setattr(self.ops, name, orig)
        self._orig = {}

    def _wrap(self, name, model):
        orig = getattr(self.ops, name)
        self._orig[name] = orig

        def timed(*a, **k):
            # (never inside a hipGraph capture: an event recorded there belongs to the graph and cannot be timed)
            m = model(a, k) if (self.on and not torch.cuda.is_current_stream_capturing()) else None
            if m is None:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.rec.setdefault(m[0], {"roc": m[1], "ev": [], "exec": m[2], "alg": m[3], "bytes": m[4], "call": name,
                                       "x6": ("x6" in m[1]) or (len(m) > 5 and bool(m[5])),
                                       "mfma": m[6] if len(m) > 6 else None})["ev"].append((e0, e1))
            return r
        setattr(self.ops, name, timed)

    def table(self, pmc=None):
        """per timed kernel: launches, mean ms (HIP events), executed FLOP per launch, fraction of the fp32 MFMA peak,
        HBM bytes per launch from the committed PMC passes (when the workload is the one they were taken on)"""
        rows = []
        for label, r in self.rec.items():
            ms = [a.elapsed_time(b) for a, b in r["ev"]]
            avg = float(np.mean(ms))
            tf = r["exec"] / (avg * 1e-3) / 1e12
            e = {"name": label, "rocprof_name": r["roc"], "call": r["call"], "x6": r["x6"], "launches_timed": len(ms), "ms": avg, "total_ms": float(np.sum(ms)),
                 "executed_flop": r["exec"], "algorithmic_flop": r["alg"], "tflops": tf, "frac": tf / PEAK_F32_TFLOPS,
                 "algorithmic_bytes": r["bytes"], "hbm_gb": None, "hbm_frac": None}
            if r.get("mfma"):      # what the matrix cores execute for it (padding and redundant products included), fp32-equivalent
                e["mfma_instr_model"] = r["mfma"]
                e["mfma_flop"] = mfma_flop(r["mfma"]) / 6.0
            rows.append(e)
        rows.sort(key=lambda e: -e["total_ms"])
        return rows


# BASELINE.json configs[1..4] at their per-GPU shard sizes: (label, alg, shape, envs per GPU, mixer dtype,
# gemm mode).  The "extra" ones are the opt-in bf16x6 legs (args.gemm_mode = "bf16x6": the agent unrolls and the QPLEX lambda-net on
# the split kernels, csrc/agent_x6.hip / mlp3_x6.hip) beside their fp32 twins; the headline and configs[1..4] stay on
# v_mfma_f32_16x16x4_f32.
OTHER_CONFIGS = [("cfg2 QMIX 2s3z 1024 envs (1 GPU)", "qmix", "2s3z", 1024, "fp32", "f32"),
                 ("cfg3 QPLEX 2s3z 512 envs (shard of 4096 / 8 GPUs)", "qplex", "2s3z", 512, "fp32", "f32"),
                 ("cfg4 QTRAN-base 3s5z 512 envs (shard of 2048 / 4 GPUs)", "qtran_base", "3s5z", 512, "fp32", "f32"),
                 ("cfg5 QMIX MMM2 1024 envs (shard of 8192 / 8 GPUs), bf16 mixer", "qmix", "MMM2", 1024, "bf16", "f32"),
                 ("extra: cfg3 shard with gemm_mode bf16x6 (agent unrolls, BPTT, lambda-net: fp32 products as six bf16 MFMA products each)", "qplex", "2s3z", 512, "fp32", "bf16x6"),
                 ("extra: QPLEX 2s3z 4096 envs on one GPU, fp32 MFMA", "qplex", "2s3z", 4096, "fp32", "f32"),
                 ("extra: QPLEX 2s3z 4096 envs on one GPU, gemm_mode bf16x6", "qplex", "2s3z", 4096, "fp32", "bf16x6"),
                 ("extra: headline learner (QMIX 2s3z 4096 envs) with gemm_mode bf16x6 (agent unrolls and BPTT on the split kernels)", "qmix", "2s3z", 4096, "fp32", "bf16x6"),
                 ("extra: cfg2 with gemm_mode bf16x6", "qmix", "2s3z", 1024, "fp32", "bf16x6"),
                 ("extra: cfg4 with gemm_mode bf16x6 (agent unrolls and BPTT on the split kernels; the QTRAN heads stay fp32)", "qtran_base", "3s5z", 512, "fp32", "bf16x6"),
                 ("extra: cfg5 with gemm_mode bf16x6 (agent unrolls and BPTT on the split kernels beside the bf16 mixer)", "qmix", "MMM2", 1024, "bf16", "bf16x6"),
                 ("extra: QMIX 2s3z 512 envs (shard of 4096 / 8 GPUs), fp32 MFMA", "qmix", "2s3z", 512, "fp32", "f32"),
                 ("extra: QMIX 2s3z 512 envs (shard of 4096 / 8 GPUs), gemm_mode bf16x6", "qmix", "2s3z", 512, "fp32", "bf16x6")]


def load_pmc(workload):
    """HBM bytes per launch from the committed PMC passes of THIS workload (profiles/<round>_pmc_<workload>.json of the newest
    round that has one, written by tools/summarize_profiles.py) - only when the file was taken with the library that is running
    now (marl_hip_version() carries a hash of the kernel sources): numbers of an older build silently go stale when a kernel
    changes."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % workload)))
    if not cands:
        return {}, "no PMC file for this workload"
    path = cands[-1]
    d = json.load(open(path))
    from marl_amd import _lib
    ver = _lib.load().marl_hip_version().decode()
    if d.get("lib_version") != ver:
        return {}, "PMC file %s is from another build (%s, running %s): not used" % (os.path.relpath(path, ROOT), d.get("lib_version"), ver)
    return d["kernels"], os.path.relpath(path, ROOT)


def pmc_traffic(pmc, e):
    """HBM bytes per launch of timed kernel entry e (None when the PMC file has no unique match for it)"""
    hit = [v for k, v in pmc.items() if k.startswith(e["rocprof_name"]) and "hbm_bytes_per_launch" in v]
    if e["rocprof_name"] == "agent_fwd":
        hit = [v for k, v in pmc.items() if k.startswith("agent_fwd") and not k.startswith("agent_fwd_x6") and "hbm_bytes_per_launch" in v]
    if e["rocprof_name"] == "agent_fwd_x6":
        # agent_fwd_x6_kernel<tiles, SAVE, XS, GIO>: match by what the launch does
        tag = e["name"].split("[")[1][:4]
        want = {"save": ("true", "false"), "reus": ("false", "true"), "plai": ("false", "false")}[tag]      # (SAVE, XS)
        hit = [v for k, v in pmc.items() if k.startswith("agent_fwd_x6_kernel<") and "hbm_bytes_per_launch" in v
               and tuple(k[k.index("<") + 1:k.rindex(">")].split(", ")[1:3]) == want]
    if e["rocprof_name"].startswith("mlp3") and len(hit) > 1:
        # one instantiation per padded input width: <8, ...> for K1 <= 128, <11, ...> (fp32) / <12, ...> (bf16x6) beyond; three-layer heads
        import re
        k1 = int(re.search(r"K1=(\d+)", e["name"]).group(1))
        kc = 8 if k1 <= 128 else (12 if "x6" in e["rocprof_name"] else 11)
        pre = "%s_kernel<%d, true" % (e["rocprof_name"], kc)
        hit = [v for k, v in pmc.items() if k.startswith(pre) and "hbm_bytes_per_launch" in v]
    if e["rocprof_name"] == "agent_fwd" and hit:
        # three instantiations share the prefix: match by what the launch does (save: most written; reuse: fewest MFMAs)
        by = sorted(hit, key=lambda v: v.get("WRITE_SIZE", 0))
        tag = e["name"].split("[")[1][:4]
        rest = by[:-1]
        key = lambda v: v.get("SQ_INSTS_MFMA", v.get("hbm_bytes_per_launch", 0))
        hit = [by[-1]] if tag == "save" else ([min(rest, key=key)] if tag == "reus" else [max(rest, key=key)]) if rest else []
    if e["rocprof_name"] == "agent_bwd_x6_kernel" and len(hit) > 1:
        # one call = up to two launches (full rounds of two-tile workgroups, then one round of one-tile ones: the <.., 2> and <.., 1>
        # instantiations of the same variant): the call's traffic is their sum
        tot = dict(hit[0])
        for k in ("hbm_bytes_per_launch", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_MFMA"):
            if all(k in h for h in hit):
                tot[k] = sum(h[k] for h in hit)
        return tot
    return hit[0] if len(hit) == 1 else None


def config_leg(label, alg, shape, envs, mixer_dtype, gemm_mode="f32", updates=12, warmup=6, ktimed=4, rewarm=10):
    """One learner-update leg of another BASELINE configuration at its per-GPU shard size (after the contract's timed
    region; record already in HBM): updates/s, transitions/s, the whole-update fraction of the fp32 MFMA peak (SURVEY 8d
    FLOP per transition) and the executed-FLOP roofline of the kernel the update spends the most time in."""
    from marl_amd import ops
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.algorithm.qtran_learner import QTRANLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.common.replaybuffer import ReplayBuffer
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    # every leg starts from a device heap without the previous legs' cached blocks and workspaces.  (The 512-env bf16x6 leg is
    # bound by the HOST's launch path - its kernels add up to 1.18 ms per update, a fresh process runs it at 805-837 updates/s, this
    # process at 650-800 depending on what ran before it with the same per-kernel times: tools/leg_seq.py.)
    ops.WS.bufs.clear()
    ops.WS.gen += 1
    gc.collect()
    torch.cuda.empty_cache()
    args = make_args(alg, shape, 0)
    args.mixer_dtype = mixer_dtype
    args.gemm_mode = gemm_mode
    args.buffer_size, args.batch_size = 2 * envs, envs
    T = args.episode_limit
    torch.manual_seed(0)
    np.random.seed(1)
    mac = SharedMAC(args)
    learner = QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args)
    env = SyntheticSMACEnv(envs, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, T, seed=1, fixed_length=True)
    worker = RolloutWorker(env, mac, args)
    # the update trains on replay-ring samples, as in the contract's step (the rollout plays into the ring; a sample is an episode
    # index into it): this is the form hipGraph replay - the default for shards of at most 1536 episodes - applies to
    buf = ReplayBuffer(args)
    worker.record_sink = buf
    timers = KernelTimers(ops, args, envs)

    try:
        for _ in range(2):
            buf.store_episode(worker.generate_episodes(envs)[0])
        ep = None
        train = lambda i: learner.train(buf.sample(envs), i)
        for i in range(warmup):
            train(i)
        torch.cuda.synchronize()
        # (a generation-2 garbage collection - the previous legs' objects - costs 35-60 ms: inside eight timed updates it
        # turned 172 updates/s into 105 on some runs)
        gc.collect()
        gc.disable()
        try:
            # rate: `updates` updates with no per-kernel events (their ~20 event records per update cost a small shard 4-15 %), three
            # such segments: the MEDIAN is the leg's rate and all three are on the line (the small shards are bound by the host's
            # launch path and a segment now and then runs 10-15 % slow with unchanged kernel times: tools/leg_seq.py)
            # the device idled through the collection above (and the leg's setup): a small shard's updates keep getting faster for
            # ~15 updates after an idle stretch (3.0 -> 2.6 ms at QPLEX 512 envs, tools/leg_updates.py), and without these untimed
            # updates the first timed segment ran 5-8 % low on every box
            for i in range(rewarm):
                train(warmup + i)
            torch.cuda.synchronize()
            warmup += rewarm
            segs = []
            for seg in range(3):
                t0 = time.perf_counter()
                for i in range(updates):
                    train(warmup + seg * updates + i)
                torch.cuda.synchronize()
                segs.append((time.perf_counter() - t0) / updates)
            dt = sorted(segs)[1]
            # ... then the kernel table from a few updates with the HIP-event timers on
            timers.on = True
            # (per-launch events need eager launches: the kernel table is taken with the graph replay off)
            graphs, learner.graphs = getattr(learner, "graphs", None), None
            for i in range(ktimed):
                train(warmup + 3 * updates + i)
            torch.cuda.synchronize()
            timers.on = False
            learner.graphs = graphs
        finally:
            gc.enable()
        r0 = time.perf_counter()
        steps = worker.generate_episodes(envs)[3]
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - r0
        kern = timers.table()
    finally:
        timers.close()
    fpt = learner_flops_per_transition(args, alg)
    upd_tf = fpt * envs * T / dt / 1e12
    workload = "%s_%s_T%d_envs%d%s%s" % (alg, shape, T, envs, "_bf16mixer" if mixer_dtype == "bf16" else "", "_bf16x6" if gemm_mode == "bf16x6" else "")
    pmc, pmc_src = load_pmc(workload)
    out = {"workload": workload, "what": label, "mixer_dtype": mixer_dtype, "gemm_mode": gemm_mode,
           "dtype": "f32 via bf16x6 split, fp32 accumulate (agent unrolls, BPTT, lambda-net heads; everything else f32)" if gemm_mode == "bf16x6" else "f32",
           "learner_updates_per_sec": 1.0 / dt, "segments_updates_per_sec": [1.0 / x for x in segs],
           "learner_transitions_per_sec": envs * T / dt, "rollout_env_steps_per_sec": steps / t_roll,
           "roofline_update": {"bound": "mfma", "flop_per_transition": fpt, "achieved": upd_tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                               "frac": upd_tf / PEAK_F32_TFLOPS},
           "kernels": [{k: e[k] for k in ("name", "launches_timed", "ms", "executed_flop", "frac")} for e in kern[:5]]}
    if kern:
        d = kern[0]
        x6k = d["x6"]
        peak = PEAK_BF16_TFLOPS / 6.0 if x6k else PEAK_F32_TFLOPS
        hit = pmc_traffic(pmc, d)
        out["roofline"] = {"bound": "mfma", "kernel": d["name"], "rocprof_name": d["rocprof_name"], "achieved": d["tflops"], "peak": peak, "unit": "TFLOP/s",
                           "frac": d["tflops"] / peak, "avg_launch_ms": d["ms"],
                           "traffic": hit["hbm_bytes_per_launch"] if hit else None,
                           "hbm_frac": (hit["hbm_bytes_per_launch"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if hit else None,
                           "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes; %s)" % pmc_src}
        if x6k:
            out["roofline"]["peak_note"] = "dense bf16 MFMA peak / 6: a bf16x6 split spends six bf16 products per fp32 product"
        busy = sum(e["total_ms"] for e in kern) / ktimed
        if busy > 1.05 * dt * 1e3:
            # small shards: the unrolls run side by side on two streams over parts of the chip (pair / chain schedule), so a
            # kernel's time is not exclusive and its fraction of the WHOLE chip's peak understates it
            out["roofline"]["note"] = ("kernels overlap on two streams at this size (%.2f ms of kernel time per %.2f ms update): "
                                       "per-kernel fractions are against the whole chip's peak" % (busy, dt * 1e3))
    mx = [e for e in kern if e["call"] == "qmix_wide_fwd"]
    if mx and mixer_dtype == "bf16":      # config 5's named roofline: the bf16 hypernet GEMM against the HBM read of the states
        m = mx[0]
        gbs = m["algorithmic_bytes"] / (m["ms"] * 1e-3) / 1e9
        hit = pmc_traffic(pmc, m)
        out["roofline_mixer"] = {"bound": "hbm", "kernel": m["name"] + " (bf16 operands)", "rocprof_name": m["rocprof_name"], "achieved": gbs,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "avg_launch_ms": m["ms"],
                                 "bytes_per_launch": m["algorithmic_bytes"], "traffic": hit["hbm_bytes_per_launch"] if hit else None,
                                 "traffic_unit": "HBM bytes per launch of the main kernel (PMC, %s)" % pmc_src}
    out["hip_graph"] = bool(graphs is not None and not graphs.disabled and graphs.replays > 0)
    del learner, worker, env, mac, ep, buf
    torch.cuda.empty_cache()
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_probe(alg, shape, T, envs, threads):
    """updates/s of the CPU oracle's learner at a given torch thread count (one untimed + two timed updates)"""
    from oracle import seeded, learners, rollout as orl
    torch.set_num_threads(threads)
    args = seeded.make_args(shape, alg, episode_limit=T)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11)
    mshapes = seeded.mixer_param_shapes(args)
    mixer = seeded.seeded_state(mshapes, seed=12) if mshapes else {}
    sy = orl.SynthSMAC(args.n_agents, args.obs_shape, args.state_shape, args.n_actions, T, seed=1)
    sy.length = lambda env, ep: np.full(len(np.atleast_1d(env)), T, dtype=np.int64)
    ep, _, _, _, _ = orl.batched_rollout(agent, args, sy, envs, 0.5, rseed=0)
    st = learners.LearnerState(args, agent, mixer)
    learners.train(st, ep, 0)
    t0 = time.time()
    for i in range(2):
        learners.train(st, ep, 1 + i)
    return 2.0 / (time.time() - t0)


def cpu_baseline(alg, shape, T, envs, budget_s, threads=0):
    """The CPU oracle (port of the reference path, pinned by the golden vectors) on a bounded sample
    of the same workload, timed on this box's host cores (torch threads = os.cpu_count()).
    threads = 0: a short sweep over 16 / 64 / all host CPUs picks the fastest thread count for the learner update
    (1 warm-up + 2 timed train() calls each; the sweep is part of the object), then 2 warm-up + >= 5 timed calls there."""
    from oracle import seeded, learners, rollout as orl
    host_cores = os.cpu_count() or 1
    args = seeded.make_args(shape, alg, episode_limit=T)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11)
    mshapes = seeded.mixer_param_shapes(args)
    mixer = seeded.seeded_state(mshapes, seed=12) if mshapes else {}
    N, O, S, A = args.n_agents, args.obs_shape, args.state_shape, args.n_actions
    sy = orl.SynthSMAC(N, O, S, A, T, seed=1)
    sy.length = lambda env, ep: np.full(len(np.atleast_1d(env)), T, dtype=np.int64)
    torch.set_num_threads(min(host_cores, 16))
    ep, _, _, steps, _ = orl.batched_rollout(agent, args, sy, envs, 0.5, rseed=0)
    sweep = {}
    if threads > 0:
        cores = threads
    else:
        # each candidate in a CHILD process under a time limit (ONE untimed + two timed updates): torch's thread pool can be
        # pathologically slow when oversubscribed, and a running update cannot be interrupted from inside
        import subprocess
        for c in sorted({min(host_cores, x) for x in (16, 64, host_cores)}):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", str(c), "--cpu-envs", str(envs), "--alg", alg,
                                    "--shape", shape, "--T", str(T)], capture_output=True, text=True, timeout=60)
                sweep[c] = float(r.stdout.strip().splitlines()[-1])
            except (subprocess.TimeoutExpired, ValueError, IndexError):
                sweep[c] = 0.0          # did not finish three updates in 60 s
        cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    st = learners.LearnerState(args, agent, mixer)
    t0 = time.time()
    ep, _, _, steps, _ = orl.batched_rollout(agent, args, sy, envs, 0.5, rseed=0)
    t_roll = time.time() - t0
    for i in range(2):                        # warm-up (allocator, thread pool, first-touch)
        learners.train(st, ep, i)
    t0 = time.time()
    reps = 0
    while reps < 5 or ((time.time() - t0) < budget_s * 0.25 and reps < 20):
        learners.train(st, ep, 2 + reps)
        reps += 1
    t_train = (time.time() - t0) / reps
    # the reference's actual serial rollout (one env, one agent at a time), a few episodes
    t0 = time.time()
    _, _, _, ssteps, _ = orl.serial_rollout(agent, args, orl.SerialSynthEnv(sy), 8, 0.5)
    t_serial = time.time() - t0
    return {"value": steps / (t_roll + t_train), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "host_cpu_count": host_cores, "cpu_model": cpu_model(), "torch_threads": cores,
            "thread_sweep_updates_per_sec": {str(k): v for k, v in sorted(sweep.items())},
            "sample": "%s %s: %d envs x T=%d batched CPU rollout + 2 warm-up and %d timed oracle train() calls on %d "
                      "torch threads (of %d host CPUs; the fastest of the swept thread counts); serial reference-style rollout of 8 episodes"
                      % (alg, shape, envs, T, reps, cores, host_cores),
            "learner_updates_per_sec": 1.0 / t_train, "learner_transitions_per_sec": envs * T / t_train,
            "batched_rollout_env_steps_per_sec": steps / t_roll, "serial_rollout_env_steps_per_sec": ssteps / t_serial}


def visible_gpus():
    """GPUs this process would see, WITHOUT touching the HIP runtime (the parent of a self-launch must stay clean: a process
    that has initialised HIP is never replaced or forked into ranks).  From the *_VISIBLE_DEVICES lists when set - the
    SMALLEST of them: HIP_ / CUDA_VISIBLE_DEVICES index into the set ROCR_VISIBLE_DEVICES leaves, so every list bounds the
    count - else from the KFD topology (nodes with a non-zero simd_count are GPUs); None when neither says."""
    counts = [len([x for x in os.environ[k].split(",") if x.strip() != ""])
              for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(k) is not None]
    if counts:
        return min(counts)
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(top):
            for line in open(os.path.join(top, d, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        return n
    except (OSError, ValueError):
        return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run this file under torch.distributed.run as a child process
    (N ranks, rendezvous on 127.0.0.1 at a port torchrun picks itself: --standalone), stdout / stderr inherited, and return
    the child's exit code."""
    import subprocess
    have = visible_gpus()
    if have is not None and have < n and os.environ.get("MARL_BENCH_ONE_DEVICE") != "1":
        print("[bench] --gpus %d asked for, %d visible on this node" % (n, have), file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n)]
    port = os.environ.get("MASTER_PORT")
    if port:
        cmd += ["--master-addr", "127.0.0.1", "--master-port", port]
    else:
        cmd += ["--standalone", "--local-addr", "127.0.0.1"]     # c10d rendezvous on a free port chosen by torchrun: no bind-then-close race
    cmd += [os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=dict(os.environ)).returncode


DTYPE_X6 = "f32 (bf16x6 split products, fp32 accumulate)"
LINE_LIMIT = 8000          # the driver keeps an 8 KB tail of stdout: the ONE final line must fit in it with room to spare


def _r(x, sig=6):
    """floats of the printed line rounded to `sig` significant digits (the full-precision line goes to bench_full.json)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(out, limit=LINE_LIMIT):
    """The ONE stdout line the driver parses, from the full result `out` (which goes to bench_full.json and stderr):
    the contract's headline fields, `config`, `roofline` of the dominant kernel with at most six kernel rows,
    `roofline_update`, `f32_mfma_twin`, `cpu_baseline`, and one short object per `configs[]` leg.  Always <= `limit` bytes:
    optional parts are dropped (least important first) until it fits."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line["config"] = _pick(out.get("config"), ("workload", "alg", "shape", "n_agents", "obs_dim", "state_dim", "n_actions", "episode_limit",
                                               "global_envs", "envs_per_gpu", "gemm_mode", "mixer_dtype", "hip_graph", "parallelism"))
    line.update(_pick(out, ("learner_updates_per_sec", "learner_transitions_per_sec", "rollout_env_steps_per_sec", "last_loss")))
    roof = out.get("roofline") or {}
    r = _pick(roof, ("bound", "kernel", "rocprof_name", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "avg_launch_ms",
                     "launches_timed", "flop_per_launch", "bytes_per_launch"))
    if isinstance(r.get("kernel"), str):
        r["kernel"] = r["kernel"][:96]
    r["kernels"] = [dict(_pick(e, ("rocprof_name", "launches_timed", "ms", "frac", "hbm_gb", "mfma_busy_frac", "mfma_instr_pmc")), name=e["name"][:64],
                         **({"mfma_instr_model": sum(e["mfma_instr_model"].values())} if e.get("mfma_instr_model") else {}))
                    for e in (roof.get("kernels") or [])[:6]]
    if "algorithmic_rate" in roof:
        r["algorithmic_tflops_unrolls"] = roof["algorithmic_rate"]["tflops"]
    line["roofline"] = r
    line["roofline_update"] = _pick(out.get("roofline_update"), ("bound", "flop_per_transition", "achieved", "peak", "unit", "frac"))
    if out.get("f32_mfma_twin"):
        tw = out["f32_mfma_twin"]
        line["f32_mfma_twin"] = dict(_pick(tw, ("value", "ms_per_step", "dtype", "learner_updates_per_sec")),
                                     roofline=_pick(tw.get("roofline"), ("rocprof_name", "frac", "avg_launch_ms", "peak")))
    if out.get("blocking_readbacks"):
        line["blocking_readbacks_ms_per_step"] = out["blocking_readbacks"]["ms_per_step"]
    line["rccl"] = _pick(out.get("rccl"), ("backend", "world_seen"))
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "host_cpu_count", "cpu_model",
                                                            "learner_updates_per_sec", "batched_rollout_env_steps_per_sec",
                                                            "serial_rollout_env_steps_per_sec"))
        line["cpu_baseline"]["sample"] = line["cpu_baseline"].get("sample", "")[:200]
    legs = []
    for c in out.get("configs") or []:
        ro = c.get("roofline") or {}
        leg = {"workload": c["workload"], "gemm_mode": c["gemm_mode"], "updates_per_sec": c["learner_updates_per_sec"],
               "segments": c.get("segments_updates_per_sec"), "roofline_update_frac": c["roofline_update"]["frac"],
               "roofline": {"kernel": ro.get("rocprof_name"), "frac": ro.get("frac"), "traffic": ro.get("traffic")}}
        if "roofline_mixer" in c:
            leg["roofline_mixer"] = _pick(c["roofline_mixer"], ("rocprof_name", "frac", "traffic", "avg_launch_ms"))
        legs.append(leg)
    if legs:
        line["configs"] = legs
    line["full"] = out.get("full", "bench_full.json")
    line = _r(line)
    # fit: drop optional detail, least important first
    drops = [lambda l: [c.pop("segments", None) for c in l.get("configs", [])],
             lambda l: l["roofline"].__setitem__("kernels", l["roofline"]["kernels"][:3]),
             lambda l: [c.pop("roofline_mixer", None) for c in l.get("configs", [])],
             lambda l: l.pop("configs", None),
             lambda l: l["roofline"].pop("kernels", None),
             lambda l: l.get("cpu_baseline", {}).pop("sample", None)]
    txt = json.dumps(line, separators=(",", ":"))
    for d in drops:
        if len(txt) <= limit:
            break
        d(line)
        txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) <= limit, len(txt)
    return txt


LINE_OUT = None              # main(): the process's real stdout (it points sys.stdout at stderr for everything but the line)


def emit(out):
    """full result -> bench_full.json beside this script (and gpurun_out/ when it exists) and stderr; the compact line -> stdout, LAST"""
    out["full"] = "bench_full.json"
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    print("[bench full] " + full, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(out), file=LINE_OUT or sys.stdout, flush=True)


class Pipeline:
    """The contract's step for one arithmetic mode: batched rollout of the rank's envs (T lock-steps, played straight into the
    replay ring) -> store -> sample -> one learner.train()."""

    def __init__(self, o, gemm_mode, rank, world, E):
        from marl_amd.controller.share_params import SharedMAC
        from marl_amd.algorithm.q_learner import QLearner
        from marl_amd.algorithm.qtran_learner import QTRANLearner
        from marl_amd.rollout import RolloutWorker
        from marl_amd.common.replaybuffer import ReplayBuffer
        from marl_amd.env.synthetic_smac import SyntheticSMACEnv
        self.o, self.E, self.world = o, E, world
        args = self.args = make_args(o.alg, o.shape, o.T)
        args.mixer_dtype = o.mixer_dtype
        args.gemm_mode = gemm_mode
        args.hip_graph = {"on": True, "off": False, "auto": None}[o.hip_graph]
        args.lazy_loss = not o.blocking_loss
        args.buffer_size = 2 * E
        args.batch_size = E
        torch.manual_seed(0)                     # identical random-init weights on every rank
        self.mac = SharedMAC(args)
        self.learner = QTRANLearner(self.mac, args) if o.alg.startswith("qtran") else QLearner(self.mac, args)
        env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1,
                               env0=rank * E, fixed_length=True)
        self.worker = RolloutWorker(env, self.mac, args)
        self.buf = ReplayBuffer(args)
        self.worker.record_sink = self.buf          # training rollouts are played straight into the replay ring
        np.random.seed(1 + rank)
        self.train_steps = 0
        self.lazy_stats = []

    def one_step(self):
        o, E = self.o, self.E
        if o.blocking_loss:
            episodes, _, _, steps = self.worker.generate_episodes(E)
        else:
            # same rollout, same device-side statistics; their copy to the host is enqueued instead of awaited (the env
            # steps are summed from the handles after the timed region's final barrier)
            episodes, st = self.worker.finish_episodes(self.worker.launch_episodes(), lazy=True)
            self.lazy_stats.append(st)
            steps = 0
        self.buf.store_episode(episodes)
        batch = self.buf.sample(min(self.buf.current_size, self.args.batch_size))
        loss = self.learner.train(batch, self.train_steps)
        self.train_steps += 1
        return steps, loss

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_region(self, timers, dev):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; the MAX over ranks of the
        time and the SUM of the env steps.  Returns (seconds, env steps, last loss)."""
        o = self.o
        for _ in range(o.warmup):
            self.one_step()
        # a full (generation-2) Python garbage collection costs ~35 ms here - three pipeline steps; collect now and
        # keep the collector off inside the timed regions (what timeit does)
        gc.collect()
        gc.disable()
        try:
            self.barrier()
            # HIP-event timing of the kernels on every THIRD step of the timed region: the ~16 event records of a step cost it 0.1 ms (1 %;
            # measured with MARL_BENCH_TIMER_STRIDE=1 / 0 on one box: 9.74 / 9.63 ms per step) - the value must not pay for its own roofline
            stride = int(os.environ.get("MARL_BENCH_TIMER_STRIDE", "3" if o.steps >= 6 else "1"))
            # small shards replay the learner's schedule as ONE hipGraph (no per-launch events inside it): their kernel table is
            # taken from two eager steps AFTER the timed region (kernel_table_steps below)
            graphs = getattr(self.learner, "graphs", None)
            self.graphed = bool(graphs is not None and not graphs.disabled and graphs.replays > 0)
            if self.graphed:
                stride = 0
            first = len(self.lazy_stats)
            t0 = time.perf_counter()
            env_steps = 0
            loss = None
            for si in range(o.steps):
                timers.on = stride > 0 and si % stride == 0
                s, loss = self.one_step()
                env_steps += s
            self.barrier()
            dt = time.perf_counter() - t0
            timers.on = False
        finally:
            gc.enable()
        env_steps += sum(st.steps() for st in self.lazy_stats[first:first + o.steps])
        if self.graphed:
            self.learner.graphs, keep = None, self.learner.graphs
            timers.on = True
            for _ in range(2):
                self.one_step()
            self.barrier()
            timers.on = False
            self.learner.graphs = keep
        if self.world > 1:
            tt = torch.tensor([dt, float(env_steps)], dtype=torch.float64, device=dev)
            tmax = tt.clone()
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            tsum = tt.clone()
            torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
            dt, env_steps = float(tmax[0]), float(tsum[1])
        return dt, env_steps, loss

    def close(self):
        self.learner = self.worker = self.buf = self.mac = None
        gc.collect()
        torch.cuda.empty_cache()


def roofline_object(kern, pmc, pmc_path, steps):
    """the roofline object of a timed region from its kernel table (KernelTimers.table): the kernel the region spent the most
    time in; `achieved` = FLOP that kernel EXECUTES per launch / its mean launch duration (HIP events inside the timed
    region).  Work that is avoided (the double-Q unroll reading the eval unroll's input-side sums) is a throughput gain and is
    NOT credited here: it shows up in `algorithmic_rate` (SURVEY 8d FLOP / time) and in the end-to-end `roofline_update`."""
    for e in kern:
        hit = pmc_traffic(pmc, e)
        if hit:
            e["hbm_gb"] = hit["hbm_bytes_per_launch"] / 1e9
            e["hbm_frac"] = hit["hbm_bytes_per_launch"] / (e["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
            if "SQ_INSTS_MFMA" in hit and not e["x6"]:
                e["mfma_flop_pmc"] = hit["SQ_INSTS_MFMA"] * 2048.0      # v_mfma_f32_16x16x4_f32: 2048 FLOP per wave-instruction
            if "SQ_INSTS_MFMA" in hit and e.get("mfma_instr_model"):
                e["mfma_instr_pmc"] = hit["SQ_INSTS_MFMA"]              # against sum(mfma_instr_model): the model's instruction count
            if "mfma_busy_frac" in hit:
                e["mfma_busy_frac"] = hit["mfma_busy_frac"]
    roof = {"bound": "mfma", "kernel": None, "achieved": None, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
    if not kern:
        return roof
    d = kern[0]
    roof.update(kernel=d["name"], rocprof_name=d["rocprof_name"], achieved=d["tflops"], frac=d["frac"],
                avg_launch_ms=d["ms"], launches_timed=d["launches_timed"], flop_per_launch=d["executed_flop"],
                traffic=(d["hbm_gb"] * 1e9 if d["hbm_gb"] else None), hbm_frac=d["hbm_frac"],
                traffic_unit="HBM bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes; %s)" % pmc_path,
                what="executed FLOP (tile padding excluded) of the kernel with the largest total time in the timed region")
    for e in kern:      # a split kernel is priced against ITS peak: six bf16 products per fp32 product
        if e["x6"]:
            e["frac"] = e["tflops"] / (PEAK_BF16_TFLOPS / 6.0)
            e["peak"] = PEAK_BF16_TFLOPS / 6.0
    roof["kernels"] = [{k: e.get(k) for k in ("name", "rocprof_name", "launches_timed", "ms", "executed_flop", "frac", "peak", "hbm_gb", "hbm_frac",
                                              "mfma_flop", "mfma_instr_model", "mfma_instr_pmc", "mfma_busy_frac")}
                       for e in kern[:6]]
    if d["x6"]:
        roof.update(peak=PEAK_BF16_TFLOPS / 6.0, frac=d["tflops"] / (PEAK_BF16_TFLOPS / 6.0),
                    peak_note="dense bf16 MFMA peak / 6: a bf16x6 split spends six bf16 products per fp32 product")
    un = [e for e in kern if e["rocprof_name"] in ("agent_fwd", "agent_fwd_x6")]
    if un:
        t_un = sum(e["total_ms"] for e in un)
        roof["algorithmic_rate"] = {
            "what": "the learner's unroll launches (three per update): SURVEY 8d FLOP (B N T F_a per launch, whether computed or "
                    "reused) / their mean duration - a throughput figure, not a pipe utilisation",
            "tflops": sum(e["algorithmic_flop"] * e["launches_timed"] for e in un) / (t_un * 1e-3) / 1e12,
            "executed_tflops": sum(e["executed_flop"] * e["launches_timed"] for e in un) / (t_un * 1e-3) / 1e12,
            "avg_launch_ms": t_un / sum(e["launches_timed"] for e in un)}
    tot_exec = sum(e["executed_flop"] * e["launches_timed"] for e in kern)
    tot_ms = sum(e["total_ms"] for e in kern)
    roof["all_timed_kernels"] = {"executed_tflops": tot_exec / (tot_ms * 1e-3) / 1e12, "ms_per_step": tot_ms / max(steps, 1)}
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="GLOBAL number of parallel envs / episodes per update")
    ap.add_argument("--alg", default="qmix")
    ap.add_argument("--shape", default="2s3z")
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--mixer-dtype", default="fp32", choices=["fp32", "bf16"], help="bf16: mixer GEMMs on the bf16 matrix cores (config 5)")
    ap.add_argument("--gemm-mode", default="bf16x6", choices=["f32", "bf16x6"],
                    help="arithmetic of the dense products.  bf16x6 (default, VERDICT r04 ruling): every fp32 operand split EXACTLY into "
                         "three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulate - fp32-accurate (same 1e-4 parity "
                         "bounds).  f32: v_mfma_f32_16x16x4_f32 everywhere; the default run times that twin too (`f32_mfma_twin`)")
    ap.add_argument("--no-twin", action="store_true", help="skip the f32 twin of the timed region (same steps, gemm_mode f32, same process)")
    ap.add_argument("--blocking-loss", "--blocking-readbacks", dest="blocking_loss", action="store_true",
                    help="read every update's loss and every rollout's statistics back at once (default: the copies are enqueued "
                         "and read at the end of the timed region - same device work, no host stall between steps)")
    ap.add_argument("--hip-graph", default="auto", choices=["auto", "on", "off"], nargs="?", const="on",
                    help="replay the learner's forward/backward schedule as one hipGraph: auto = below ~1500 envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the learner legs of the other BASELINE configurations (configs[1..4] at "
                    "their per-GPU shard sizes) that the default single-GPU run appends to the line as `configs`")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-probe", type=int, default=0, help=argparse.SUPPRESS)     # child mode of the CPU thread sweep
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline (0: sweep 16 / 64 / all host CPUs, keep the fastest)")
    ap.add_argument("--leg-iters", type=int, default=5, help="iterations of the separately timed learner / rollout legs")
    ap.add_argument("--roofline-kernel", default="unroll", choices=["unroll", "mixer"],
                    help="kernel the roofline object describes: the dominant kernel of the timed region (MFMA bound; headline) or the fused "
                         "wide-state QMIX forward (config 5: HBM bound on reading the states when --mixer-dtype bf16)")
    ap.add_argument("--dry", action="store_true", help="multi-GPU pre-flight only: init RCCL, one all-reduce of the real "
                    "gradient-buffer size, print the result and exit")
    o = ap.parse_args()
    if o.cpu_probe:
        print(cpu_probe(o.alg, o.shape, o.T or SHAPES[o.shape][4], o.cpu_envs, o.cpu_probe))
        return

    if o.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: nothing in this process has touched the GPU yet (importing torch does not, and the
        # GPUs are counted from the environment / the KFD topology), so start the N ranks as a FRESH child - torch.distributed.run,
        # one process per GPU - let it print rank 0's JSON line on our stdout and leave with its return code (a process that has
        # initialised HIP must never be replaced by exec)
        sys.exit(self_launch(o.gpus))

    # stdout carries the ONE JSON line and nothing else: what the classes print as the reference's do ('Init RolloutWorker',
    # rollout.py:135) goes to stderr from here on
    global LINE_OUT
    LINE_OUT = sys.stdout
    sys.stdout = sys.stderr
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if os.environ.get("MARL_BENCH_ONE_DEVICE") == "1":      # test mode: every rank on GPU 0 (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from marl_amd.hostutil import pin_to_gpu_numa
    numa = pin_to_gpu_numa(local)            # one process per GPU, on the CPUs of that GPU's NUMA node (two-socket hosts)
    from marl_amd import experiments
    forced = world == 1 and experiments.get("force_reducer") == 1
    if forced:      # MARL_FORCE_REDUCER=1 on one GPU: a world-1 RCCL group, so every update pays its collectives (what they cost at N = 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        import torch.distributed as dist
        backend = os.environ.get("MARL_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == o.gpus, "launch with torch.distributed.run --nproc-per-node %d" % o.gpus
    if world > 1 or o.dry:
        # pre-flight: the exchange step of an update (one flat fp32 all-reduce, one int32 MAX all-reduce, one broadcast)
        # BEFORE anything is built, so a broken RCCL / IPC setup fails here, fast and legibly
        import torch.distributed as dist
        if world > 1:
            t0 = time.perf_counter()
            probe = torch.full((70000,), float(rank + 1), device=dev)         # ~ the QMIX-2s3z gradient buffer (62 896 floats)
            dist.all_reduce(probe)
            ti = torch.tensor([rank + 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ti, op=dist.ReduceOp.MAX)
            dist.broadcast(probe, src=0)
            torch.cuda.synchronize()
            ok = bool(abs(float(probe[0]) - world * (world + 1) / 2) < 1e-3 and int(ti) == world)
            if rank == 0:
                print("[bench preflight] backend=%s world=%d all_reduce/max/broadcast %s in %.1f ms (HSA_ENABLE_IPC_MODE_LEGACY=%s)"
                      % (dist.get_backend(), world, "OK" if ok else "WRONG RESULT", (time.perf_counter() - t0) * 1e3,
                         os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")), file=sys.stderr, flush=True)
            assert ok, "RCCL pre-flight returned wrong values"
        if o.dry:
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            elif rank == 0:
                print("[bench preflight] single process: nothing to check", file=sys.stderr)
            return

    from marl_amd import ops
    E = o.envs // world                      # envs / episodes per rank
    pipe = Pipeline(o, o.gemm_mode, rank, world, E)
    args, learner, worker, buf = pipe.args, pipe.learner, pipe.worker, pipe.buf
    T, N = args.episode_limit, args.n_agents
    barrier = pipe.barrier

    # HIP-event timing of every heavy C-ABI call inside the timed region (on the stream it is launched on)
    timers = KernelTimers(ops, args, E)
    dt, env_steps, loss = pipe.timed_region(timers, dev)

    # separately timed legs (after the contract's timed region): learner-only and rollout-only, three passes each, the MEDIAN
    # counts and all three are in the full result
    gc.collect()
    gc.disable()
    batch = buf.sample(E)
    learn_p, roll_p = [], []
    for rep in range(3):
        barrier(); t1 = time.perf_counter()
        for i in range(o.leg_iters):
            learner.train(batch, 10 ** 6 + rep * o.leg_iters + i)
        barrier(); learn_p.append((time.perf_counter() - t1) / o.leg_iters)
    for rep in range(3):
        barrier(); t1 = time.perf_counter()
        rs = 0
        for i in range(o.leg_iters):
            rs += worker.generate_episodes(E)[3]
        barrier(); roll_p.append((time.perf_counter() - t1) / o.leg_iters)
    t_learn, t_roll = sorted(learn_p)[1], sorted(roll_p)[1]
    # the same pipeline step with the REFERENCE's host semantics: every rollout's statistics and every update's loss are read
    # back at once (runner.py:85-98 uses both immediately); same device work, the host waits for it twice per step
    t_block = None
    if not o.blocking_loss:
        args.lazy_loss = False
        learner.loss_readback.lazy = False
        blk_steps = 0
        barrier(); t1 = time.perf_counter()
        for i in range(o.leg_iters):
            episodes, _, _, st_ = worker.generate_episodes(E)
            buf.store_episode(episodes)
            float(learner.train(buf.sample(min(buf.current_size, args.batch_size)), pipe.train_steps))
            pipe.train_steps += 1
            blk_steps += st_
        barrier(); t_block = (time.perf_counter() - t1) / o.leg_iters
        blk_rate = blk_steps / (t_block * o.leg_iters)
    gc.enable()
    graph_on = pipe.graphed
    kern = timers.table()
    timers.close()
    del learner, worker, buf, batch
    pipe.close()

    # the f32 twin: the SAME timed region (same steps, warm-up, envs, seeds) with every product on v_mfma_f32_16x16x4_f32, in this
    # process, so that the fp32-MFMA number stays on the line beside the split-mode headline
    twin = None
    if o.gemm_mode == "bf16x6" and not o.no_twin:
        tp = Pipeline(o, "f32", rank, world, E)
        ttimers = KernelTimers(ops, tp.args, E)
        tdt, tsteps, tloss = tp.timed_region(ttimers, dev)
        tkern = ttimers.table()
        ttimers.close()
        gc.collect(); gc.disable()
        tb = tp.buf.sample(E)
        tl = []
        for rep in range(3):
            tp.barrier(); t1 = time.perf_counter()
            for i in range(o.leg_iters):
                tp.learner.train(tb, 10 ** 6 + rep * o.leg_iters + i)
            tp.barrier(); tl.append((time.perf_counter() - t1) / o.leg_iters)
        gc.enable()
        del tb
        tp.close()
        if rank == 0:
            wl = "%s_%s_T%d_envs%d%s" % (o.alg, o.shape, T, E, "_bf16mixer" if o.mixer_dtype == "bf16" else "")
            tpmc, tpath = load_pmc(wl) if world == 1 else ({}, "multi-GPU run: per-GPU shard")
            twin = {"what": "the same timed region with gemm_mode f32 (every product on v_mfma_f32_16x16x4_f32), same process",
                    "value": tsteps / tdt, "unit": "env-steps/s", "ms_per_step": tdt / o.steps * 1e3, "dtype": "f32",
                    "last_loss": float(tloss), "learner_updates_per_sec": 1.0 / sorted(tl)[1],
                    "learner_passes_updates_per_sec": [1.0 / x for x in tl],
                    "roofline": roofline_object(tkern, tpmc, tpath, o.steps)}

    if rank == 0:
        fpt = learner_flops_per_transition(args, o.alg)
        upd_tflops = fpt * (o.envs * T / t_learn) / 1e12
        # HBM bytes per launch from the committed PMC passes of this workload, taken with this very library (load_pmc)
        workload = "%s_%s_T%d_envs%d%s%s" % (o.alg, o.shape, T, o.envs // world, "_bf16mixer" if o.mixer_dtype == "bf16" else "",
                                             "_bf16x6" if o.gemm_mode == "bf16x6" else "")
        pmc, pmc_path = load_pmc(workload) if world == 1 else ({}, "multi-GPU run: per-GPU shard")
        # (with hipGraph replay the learner's kernels are launched from inside the replayed graph: no per-launch events)
        roof = roofline_object(kern, pmc, pmc_path, o.steps)
        if o.roofline_kernel == "mixer":
            # fused wide-state QMIX forward: one launch reads every state row once (4 S bytes), the chosen Qs (4 N) and
            # writes q_tot (4): algorithmic bytes = rows * (4 S + 4 N + 4), rows = envs per GPU * T (with bf16
            # operands the hypernet GEMM sits below the bf16 ridge, i.e. it is bound by this read)
            mx = [e for e in kern if e["call"] == "qmix_wide_fwd"]
            if mx:
                m = mx[0]
                gbs = m["algorithmic_bytes"] / (m["ms"] * 1e-3) / 1e9
                hit = pmc_traffic(pmc, m)
                roof = {"bound": "hbm", "kernel": m["name"] + " (%s operands)" % o.mixer_dtype, "rocprof_name": m["rocprof_name"],
                        "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "traffic": hit["hbm_bytes_per_launch"] if hit else None, "avg_launch_ms": m["ms"], "launches_timed": m["launches_timed"], "bytes_per_launch": m["algorithmic_bytes"],
                        "flop_per_launch": m["executed_flop"], "tflops": m["tflops"], "kernels": roof.get("kernels")}
        rccl = {"backend": "none (single process)", "world_seen": 1}
        if world > 1:
            rccl = {"backend": torch.distributed.get_backend(), "world_seen": torch.distributed.get_world_size(),
                    "preflight": "all_reduce(sum) / all_reduce(max) / broadcast verified on every rank before the run"}
        out = {
            "metric": "env_steps_per_sec", "value": env_steps / dt, "unit": "env-steps/s",
            "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "ms_per_step": dt / o.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if o.gemm_mode == "f32" else DTYPE_X6,
            "data": "synthetic",
            "config": {"workload": "%s_%s_T%d_envs%d" % (o.alg, o.shape, T, o.envs), "alg": o.alg, "shape": o.shape,
                       "n_agents": N, "obs_dim": args.obs_shape, "state_dim": args.state_shape,
                       "n_actions": args.n_actions, "episode_limit": T, "global_envs": o.envs, "envs_per_gpu": E,
                       "gemm_mode": o.gemm_mode, "mixer_dtype": o.mixer_dtype, "hip_graph": graph_on,
                       "kernel_table": "two eager steps after the timed region (its steps replay the learner's schedule as one hipGraph)" if graph_on
                                       else "HIP events inside the timed region",
                       "parallelism": "dp%d" % world, "numa_node": numa,
                       "step": "batched rollout (T lock-steps) + replay store/sample + 1 learner.train()"},
            "learner_updates_per_sec": 1.0 / t_learn, "learner_passes_updates_per_sec": [1.0 / x for x in learn_p],
            "learner_transitions_per_sec": o.envs * T / t_learn,
            "rollout_env_steps_per_sec": rs * world / o.leg_iters / t_roll,
            "rollout_passes_env_steps_per_sec": [rs * world / o.leg_iters / x for x in roll_p],
            "last_loss": float(loss), "loss_readback": "blocking" if o.blocking_loss else "deferred",
            "blocking_readbacks": None if t_block is None else {
                "what": "the same step with the reference's host semantics (loss and rollout statistics read back every step); "
                        "per-rank figure of rank 0, after the timed region", "ms_per_step": t_block * 1e3,
                "env_steps_per_sec_per_gpu": blk_rate},
            "roofline": roof, "rccl": rccl,
            "roofline_update": {"bound": "mfma", "what": "whole learner update (all kernels, host gaps included) against the fp32 MFMA peak: "
                                "in bf16x6 mode a throughput figure in fp32-equivalent FLOP, it may exceed 1",
                                "flop_per_transition": fpt, "achieved": upd_tflops, "peak": PEAK_F32_TFLOPS,
                                "unit": "TFLOP/s", "frac": upd_tflops / PEAK_F32_TFLOPS},
            "f32_mfma_twin": twin,
        }
        if not o.no_configs and world == 1 and (o.alg, o.shape, o.envs) == ("qmix", "2s3z", 4096):
            # the other BASELINE configurations, each at its per-GPU shard size, on this one GPU (legs after the timed region)
            out["configs"] = [config_leg(*c) for c in OTHER_CONFIGS]
        if not o.no_cpu_baseline and world == 1:      # a reported baseline of the N=1 line only
            out["cpu_baseline"] = cpu_baseline(o.alg, o.shape, T, o.cpu_envs, budget_s=20, threads=o.cpu_threads)
        emit(out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
