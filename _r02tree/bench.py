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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(alg, shape, T, envs, budget_s):
    """The CPU oracle (port of the reference path, pinned by the golden vectors) on a bounded sample
    of the same workload, timed on this box's host cores: 2 warm-up + 5 timed train() calls."""
    from oracle import seeded, learners, rollout as orl
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)     # torch-CPU oversubscribes beyond ~16 threads on these small ops (stated below)
    torch.set_num_threads(cores)
    args = seeded.make_args(shape, alg, episode_limit=T)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11)
    mshapes = seeded.mixer_param_shapes(args)
    mixer = seeded.seeded_state(mshapes, seed=12) if mshapes else {}
    st = learners.LearnerState(args, agent, mixer)
    N, O, S, A = args.n_agents, args.obs_shape, args.state_shape, args.n_actions
    sy = orl.SynthSMAC(N, O, S, A, T, seed=1)
    sy.length = lambda env, ep: np.full(len(np.atleast_1d(env)), T, dtype=np.int64)
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
            "sample": "%s %s: %d envs x T=%d batched CPU rollout + 2 warm-up and %d timed oracle train() calls on %d "
                      "torch threads (of %d host CPUs); serial reference-style rollout of 8 episodes"
                      % (alg, shape, envs, T, reps, cores, host_cores),
            "learner_updates_per_sec": 1.0 / t_train, "learner_transitions_per_sec": envs * T / t_train,
            "batched_rollout_env_steps_per_sec": steps / t_roll, "serial_rollout_env_steps_per_sec": ssteps / t_serial}


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
    ap.add_argument("--blocking-loss", "--blocking-readbacks", dest="blocking_loss", action="store_true",
                    help="read every update's loss and every rollout's statistics back at once (default: the copies are enqueued "
                         "and read at the end of the timed region - same device work, no host stall between steps)")
    ap.add_argument("--hip-graph", action="store_true", help="replay the learner's forward/backward schedule as one hipGraph (opt-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--leg-iters", type=int, default=5, help="iterations of the separately timed learner / rollout legs")
    ap.add_argument("--roofline-kernel", default="unroll", choices=["unroll", "mixer"],
                    help="kernel the roofline object describes: the agent unroll (fp32 MFMA bound; headline) or the fused "
                         "wide-state QMIX forward (config 5: HBM bound on reading the states when --mixer-dtype bf16)")
    ap.add_argument("--dry", action="store_true", help="multi-GPU pre-flight only: init RCCL, one all-reduce of the real "
                    "gradient-buffer size, print the result and exit")
    o = ap.parse_args()

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
    if world > 1:
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
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.algorithm.qtran_learner import QTRANLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.common.replaybuffer import ReplayBuffer
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv

    args = make_args(o.alg, o.shape, o.T)
    args.mixer_dtype = o.mixer_dtype
    args.hip_graph = o.hip_graph
    args.lazy_loss = not o.blocking_loss
    T, N = args.episode_limit, args.n_agents
    E = o.envs // world                      # envs / episodes per rank
    args.buffer_size = 2 * E
    args.batch_size = E
    torch.manual_seed(0)                     # identical random-init weights on every rank
    mac = SharedMAC(args)
    learner = QTRANLearner(mac, args) if o.alg.startswith("qtran") else QLearner(mac, args)
    env = SyntheticSMACEnv(E, N, args.obs_shape, args.state_shape, args.n_actions, T, seed=1, env0=rank * E,
                           fixed_length=True)
    worker = RolloutWorker(env, mac, args)
    buf = ReplayBuffer(args)
    worker.record_sink = buf          # training rollouts are played straight into the replay ring
    np.random.seed(1 + rank)

    # HIP-event timing of the dominant kernel (the persistent agent unroll, 3 launches per update)
    ev_pairs, xs_pairs = [], []      # launches doing the full algorithmic work / the double-Q launch that reuses fc1 outputs
    orig_fwd = ops.agent_unroll_fwd
    timing = {"on": False}

    def timed_fwd(*a, **k):
        if timing["on"] and a[13] > 1:       # T > 1: learner unrolls only
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_fwd(*a, **k)
            e1.record()
            (xs_pairs if k.get("gi_in") is not None else ev_pairs).append((e0, e1))
        else:
            orig_fwd(*a, **k)
    ops.agent_unroll_fwd = timed_fwd
    mix_pairs = []
    orig_wide = ops.qmix_wide_fwd

    def timed_wide(*a, **k):
        if timing["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_wide(*a, **k)
            e1.record()
            mix_pairs.append((e0, e1))
        else:
            orig_wide(*a, **k)
    ops.qmix_wide_fwd = timed_wide

    train_steps = [0]

    lazy_stats = []

    def one_step():
        if o.blocking_loss:
            episodes, _, _, steps = worker.generate_episodes(E)
        else:
            # same rollout, same device-side statistics; their copy to the host is enqueued instead of awaited (the env
            # steps are summed from the handles after the timed region's final barrier)
            episodes, st = worker.finish_episodes(worker.launch_episodes(), lazy=True)
            lazy_stats.append(st)
            steps = 0
        buf.store_episode(episodes)
        batch = buf.sample(min(buf.current_size, args.batch_size))
        loss = learner.train(batch, train_steps[0])
        train_steps[0] += 1
        return steps, loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(o.warmup):
        one_step()
    # a full (generation-2) Python garbage collection costs ~35 ms here - three pipeline steps; collect now and
    # keep the collector off inside the timed regions (what timeit does)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    timing["on"] = True
    t0 = time.perf_counter()
    env_steps = 0
    for _ in range(o.steps):
        ts = time.perf_counter()
        s, loss = one_step()
        env_steps += s
        if os.environ.get("MARL_BENCH_DEBUG"):
            torch.cuda.synchronize()
            import gc
            ms = torch.cuda.memory_stats()
            print("step %.2f ms gc=%s segs=%d reserved=%.2fGB allocs=%d" % ((time.perf_counter() - ts) * 1e3, gc.get_count(),
                  ms["segment.all.current"], ms["reserved_bytes.all.current"] / 2**30, ms["allocation.all.allocated"]), file=sys.stderr)
    barrier()
    dt = time.perf_counter() - t0
    timing["on"] = False
    env_steps += sum(st.steps() for st in lazy_stats[o.warmup:o.warmup + o.steps])
    tt = torch.tensor([dt, float(env_steps)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tsum = tt.clone()
        torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        dt, env_steps = float(tmax[0]), float(tsum[1])
    kernel_ms = [a.elapsed_time(b) for a, b in ev_pairs]
    xs_ms = [a.elapsed_time(b) for a, b in xs_pairs]

    # separately timed legs (after the contract's timed region): learner-only and rollout-only
    batch = buf.sample(E)
    barrier(); t1 = time.perf_counter()
    for i in range(o.leg_iters):
        learner.train(batch, 10 ** 6 + i)
    barrier(); t_learn = (time.perf_counter() - t1) / o.leg_iters
    barrier(); t1 = time.perf_counter()
    rs = 0
    for i in range(o.leg_iters):
        rs += worker.generate_episodes(E)[3]
    barrier(); t_roll = (time.perf_counter() - t1) / o.leg_iters
    gc.enable()

    if rank == 0:
        fl = agent_flops(args) * E * N * T                 # algorithmic FLOP of one unroll launch
        traffic = None                                     # HBM bytes/launch from the committed PMC passes (same workload only)
        pmc = next((p for p in (os.path.join(ROOT, "profiles", n) for n in ("r02_pmc.json", "r01_pmc.json")) if os.path.exists(p)), "")
        if pmc and (o.alg, o.shape, o.envs, world, T) == ("qmix", "2s3z", 4096, 1, 120):
            ks = [v for k, v in json.load(open(pmc))["kernels"].items() if k.startswith("agent_fwd_kernel")]
            n = sum(v["launches"] for v in ks)
            if n and all("hbm_bytes_per_launch" in v for v in ks):
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / n
        # (with --hip-graph the unrolls are launched from inside the replayed graph: no per-launch events, fields null)
        # `achieved` follows the contract: ALGORITHMIC FLOP per launch (B*N*T*F_a) over the mean duration of the
        # learner's unroll launches in the timed region (three per update).  One of the three - the double-Q unroll - reads
        # the input-side work (fc1, x W_ih) the eval unroll stored instead of repeating it, and the eval unroll pays for
        # those stores; `executed` says what the matrix pipe really did over the same launches, `by_launch` times each kind.
        all_ms = kernel_ms + xs_ms
        avg_ms = float(np.mean(all_ms)) if all_ms else None
        ach = fl / (avg_ms * 1e-3) / 1e12 if avg_ms else None
        fpt = learner_flops_per_transition(args, o.alg)
        upd_tflops = fpt * (o.envs * T / t_learn) / 1e12
        roof = {"bound": "mfma", "kernel": "agent_fwd_kernel (persistent GRU unroll, fp32 MFMA 16x16x4)",
                "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_TFLOPS if ach else None,
                "hbm_frac": (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if (traffic and avg_ms) else None,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, "
                "separate passes; %s)" % (os.path.relpath(pmc, ROOT) if pmc else "no PMC file for this workload"),
                "avg_launch_ms": avg_ms, "launches_timed": len(all_ms), "flop_per_launch": fl}
        if all_ms:
            I_ = args.obs_shape + args.n_actions + N
            # fc1 and the input-side gate products (x W_ih) of all steps but the last are read, not computed, in a reuse launch
            fl_x = fl - (2 * I_ * args.rnn_hidden_dim + 6 * args.rnn_hidden_dim ** 2) * E * N * (T - 1)
            ex = (fl * len(kernel_ms) + fl_x * len(xs_ms)) / (sum(all_ms) * 1e-3) / 1e12
            roof["executed"] = {"achieved": ex, "frac": ex / PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                                "what": "FLOP the launches executed (reuse launches: fc1 and x W_ih of all steps but the last are loaded)"}
            roof["by_launch"] = {"full": {"avg_launch_ms": float(np.mean(kernel_ms)) if kernel_ms else None, "launches_timed": len(kernel_ms),
                                          "flop_executed": fl, "what": "eval current-Q (saving activations + gate sums) and target next-Q"},
                                 "reuse": {"avg_launch_ms": float(np.mean(xs_ms)) if xs_ms else None, "launches_timed": len(xs_ms),
                                           "flop_executed": fl_x, "what": "double-Q unroll reading the eval unroll's input-side gate sums"}}
        if o.roofline_kernel == "mixer":
            # fused wide-state QMIX forward: one launch reads every state row once (4 S bytes), the chosen Qs (4 N) and
            # writes q_tot (4): algorithmic bytes = rows * (4 S + 4 N + 4), rows = envs per GPU * T (with bf16
            # operands the hypernet GEMM sits below the bf16 ridge, i.e. it is bound by this read)
            mix_ms = [a_.elapsed_time(b_) for a_, b_ in mix_pairs]
            rows_l = E * T
            byts = rows_l * (4 * args.state_shape + 4 * N + 4)
            m_ms = float(np.mean(mix_ms)) if mix_ms else None
            gbs = byts / (m_ms * 1e-3) / 1e9 if m_ms else None
            flm = 2 * args.state_shape * (N * args.qmix_hidden_dim + 3 * args.qmix_hidden_dim) * rows_l
            roof = {"bound": "hbm", "kernel": "qmix_wide_kernel forward (hypernet GEMM + mixing, %s operands)" % o.mixer_dtype,
                    "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS if gbs else None,
                    "traffic": None, "avg_launch_ms": m_ms, "launches_timed": len(mix_ms), "bytes_per_launch": byts,
                    "flop_per_launch": flm, "tflops": flm / (m_ms * 1e-3) / 1e12 if m_ms else None}
        out = {
            "metric": "env_steps_per_sec", "value": env_steps / dt, "unit": "env-steps/s",
            "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "ms_per_step": dt / o.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s_%s_T%d_envs%d" % (o.alg, o.shape, T, o.envs), "alg": o.alg, "shape": o.shape,
                       "n_agents": N, "obs_dim": args.obs_shape, "state_dim": args.state_shape,
                       "n_actions": args.n_actions, "episode_limit": T, "global_envs": o.envs, "envs_per_gpu": E,
                       "mixer_dtype": o.mixer_dtype, "hip_graph": bool(o.hip_graph),
                       "parallelism": "dp%d" % world, "numa_node": numa,
                       "step": "batched rollout (T lock-steps) + replay store/sample + 1 learner.train()"},
            "learner_updates_per_sec": 1.0 / t_learn,
            "learner_transitions_per_sec": o.envs * T / t_learn,
            "rollout_env_steps_per_sec": rs * world / o.leg_iters / t_roll,
            "last_loss": float(loss), "loss_readback": "blocking" if o.blocking_loss else "deferred",
            "roofline": roof,
            "roofline_update": {"bound": "mfma", "what": "whole learner update (all kernels, host gaps included)",
                                "flop_per_transition": fpt, "achieved": upd_tflops, "peak": PEAK_F32_TFLOPS,
                                "unit": "TFLOP/s", "frac": upd_tflops / PEAK_F32_TFLOPS},
        }
        if not o.no_cpu_baseline and world == 1:      # a reported baseline of the N=1 line only
            out["cpu_baseline"] = cpu_baseline(o.alg, o.shape, T, o.cpu_envs, budget_s=20)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
