#!/usr/bin/env python3
"""Same-process A/B of library debug settings on the bench pipeline step (rollout -> store -> sample -> train), alternating:
    python tools/ab_steps.py 512 gip=1,rt=1 gip=0,rt=1 gip=1,rt=2 [--alg qmix --shape 2s3z --reps 4 --steps 30]
settings: gip = marl_debug_set_fwd_gip, rt = marl_debug_set_pipe_max_rt, brt = marl_debug_set_bwd_pipe_max_rt (when exported).
Prints ms per step (wall clock over `steps` steps between synchronisations) per setting and repetition, and the learner-only
and rollout-only times."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser()
ap.add_argument("envs", type=int)
ap.add_argument("settings", nargs="+")
ap.add_argument("--alg", default="qmix"); ap.add_argument("--shape", default="2s3z")
ap.add_argument("--reps", type=int, default=4); ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--mixer-dtype", default="fp32")
o = ap.parse_args()
from marl_amd import _lib
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.algorithm.qtran_learner import QTRANLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.common.replaybuffer import ReplayBuffer
lib = _lib.load()
SET = {"gip": "marl_debug_set_fwd_gip", "rt": "marl_debug_set_pipe_max_rt", "brt": "marl_debug_set_bwd_pipe_max_rt",
       "rgate": "marl_debug_set_rollout_split"}
def apply(s):
    for kv in s.split(","):
        k, v = kv.split("=")
        getattr(lib, SET[k])(int(v))
E = o.envs
args = bench.make_args(o.alg, o.shape, 0); args.buffer_size = 2 * E; args.batch_size = E; args.mixer_dtype = o.mixer_dtype
args.lazy_loss = True
torch.manual_seed(0)
mac = SharedMAC(args); learner = QTRANLearner(mac, args) if o.alg.startswith("qtran") else QLearner(mac, args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args); buf = ReplayBuffer(args); w.record_sink = buf
ts = [0]
def step():
    ep, st = w.finish_episodes(w.launch_episodes(), lazy=True)
    buf.store_episode(ep)
    learner.train(buf.sample(min(buf.current_size, args.batch_size)), ts[0]); ts[0] += 1
def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
import gc
for s in o.settings:
    apply(s)
    for _ in range(4): step()
gc.collect(); gc.disable()
res = {s: [] for s in o.settings}
for r in range(o.reps):
    for s in o.settings:
        apply(s)
        for _ in range(2): step()
        t_step = timed(step, o.steps)
        b = buf.sample(E)
        t_learn = timed(lambda: learner.train(b, 10**6), max(5, o.steps // 3))
        t_roll = timed(lambda: w.finish_episodes(w.launch_episodes(), lazy=True), max(5, o.steps // 3))
        res[s].append((t_step, t_learn, t_roll))
for s in o.settings:
    v = res[s]
    print("%-22s step %s | learner %s | rollout %s ms  (min step %.3f learn %.3f roll %.3f)" % (s, " ".join("%.3f" % x[0] for x in v),
          " ".join("%.3f" % x[1] for x in v), " ".join("%.3f" % x[2] for x in v), min(x[0] for x in v), min(x[1] for x in v), min(x[2] for x in v)))
