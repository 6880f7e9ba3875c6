import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from marl_amd.controller.share_params import SharedMAC
from marl_amd.algorithm.q_learner import QLearner
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.common.replaybuffer import ReplayBuffer
E = int(sys.argv[1])
if os.environ.get("MARL_PIPE_MAX_RT"):
    from marl_amd import _lib
    _lib.load().marl_debug_set_pipe_max_rt(int(os.environ["MARL_PIPE_MAX_RT"]))
args = bench.make_args("qmix", "2s3z", 0); args.buffer_size = 2 * E; args.batch_size = E
mac = SharedMAC(args); learner = QLearner(mac, args)
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args); buf = ReplayBuffer(args); w.record_sink = buf
def step(i):
    ep = w.generate_episodes(E)[0]; buf.store_episode(ep); b = buf.sample(min(buf.current_size, args.batch_size)); learner.train(b, i)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(38)
