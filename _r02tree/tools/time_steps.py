#!/usr/bin/env python3
"""Per-step wall time of the bench pipeline (rollout -> store -> sample -> train), with leg split."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.common.replaybuffer import ReplayBuffer
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if os.environ.get("MARL_PIPE_MAX_RT"):
    from marl_amd import _lib
    _lib.load().marl_debug_set_pipe_max_rt(int(os.environ["MARL_PIPE_MAX_RT"]))
args = bench.make_args("qmix", "2s3z", 0); args.buffer_size = 2 * E; args.batch_size = E
mac = SharedMAC(args); learner = QLearner(mac, args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args); buf = ReplayBuffer(args); w.record_sink = buf
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for i in range(10):
    t0 = sync(); ep = w.generate_episodes(E)[0]
    t1 = sync(); buf.store_episode(ep)
    t2 = sync(); b = buf.sample(min(buf.current_size, args.batch_size))
    t3 = sync(); learner.train(b, i)
    t4 = sync()
    print("step %d rollout %.2f store %.2f sample %.2f train %.2f total %.2f ms" % (i, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3))
