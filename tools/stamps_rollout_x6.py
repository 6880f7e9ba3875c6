#!/usr/bin/env python3
"""Where a lock-step of the split rollout kernel (csrc/rollout_x6.hip) goes, per wave of workgroup 0 (diagnostic build
`make -C marl_amd/csrc stamps`):  python tools/stamps_rollout_x6.py [envs]
team R (waves 0-3): A recurrence | B1 | B observations of slot t+2 (first part) | B2 | C rest + state + availability | B3
team I (waves 4-7): A fc1 + hashes | B1 | B fc2 + choice | B2 | C x + env step | B3
MARL_ROLLOUT_V1=1: the round-5 kernel (P1 | B1 | P2 | B2 | P3 | B3 | P4 | B4)"""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MARL_HIP_LIB", os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so"))      # (a variant build: MARL_HIP_LIB=marl_amd/variants/...)
sys.path.insert(0, HERE)
import torch  # noqa: E402
from marl_amd import _lib  # noqa: E402
from stamps import show  # noqa: E402
import bench  # noqa: E402
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
buf = torch.zeros(16 * 16, dtype=torch.int64, device="cuda")
V1 = os.environ.get("MARL_ROLLOUT_V1") == "1"
if not V1:
    from marl_amd import experiments
    experiments.set("rollout_v1", 2)         # force this round's kernel at every batch size
fn = lib.marl_debug_stamps_rollout_x6_v1 if V1 else lib.marl_debug_stamps_rollout_x6
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf.data_ptr()) == 0
from marl_amd.controller.share_params import SharedMAC
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
args = bench.make_args("qmix", "2s3z", 0)
args.gemm_mode = "bf16x6"
mac = SharedMAC(args); mac.cuda()
env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
w = RolloutWorker(env, mac, args)
for _ in range(2):
    buf.zero_()
    w.generate_episodes(E)
    torch.cuda.synchronize()
show(buf.cpu().view(16, 16).numpy(), ["P1", "B1", "P2", "B2", "P3", "B3", "P4", "B4"] if V1 else ["A", "B1", "B(choice)", "B2", "C(env)", "B3", "B(q)", "C(x)"],
     "split rollout" + (" (round-5 kernel)" if V1 else ""), E, args.episode_limit)
