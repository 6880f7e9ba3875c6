#!/usr/bin/env python3
"""In-kernel segment timing (diagnostic build `make -C marl_amd/csrc stamps`): prints, per wave of workgroup 0,
the share of s_memtime cycles each stamped segment of a kernel took.  Shares only - the stamped build is slower."""
import os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MARL_HIP_LIB"] = os.path.join(HERE, "marl_amd", "libmarl_hip_stamps.so")
sys.path.insert(0, HERE)
import torch  # noqa: E402
import bench  # noqa: E402
from marl_amd import _lib  # noqa: E402

SEGS = {
    "rollout": ["fc1", "bar1", "gen_slot", "gru", "bar2", "fc2", "bar3", "choice", "bar4", "envstep"],
    "fwd": ["fc1", "bar1", "commit", "gru", "bar2", "fc2"],
    "fwd_pipe": ["side", "gru", "commit", "bar"],
    "qmix": ["top+fetch", "bar1", "mfma", "pa", "bar2", "finish", "bwd", "stash"],
    "wgrad": ["barrier", "work"],
    "bwd_pipe": ["product", "gates/dxp", "accum", "barrier"],
    "bwd": ["phaseB", "bar1", "phaseC", "dqwrite", "bar2"],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "rollout"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    lib = _lib.load()
    buf = torch.zeros(16 * 16, dtype=torch.int64, device="cuda")
    import ctypes
    fn = getattr(lib, "marl_debug_stamps_" + which.replace("_pipe", ""))
    fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
    assert fn(buf.data_ptr()) == 0
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.rollout import RolloutWorker
    from marl_amd.env.synthetic_smac import SyntheticSMACEnv
    args = bench.make_args("qmix", "2s3z", 0)
    mac = SharedMAC(args)
    env = SyntheticSMACEnv(E, args.n_agents, args.obs_shape, args.state_shape, args.n_actions, args.episode_limit, seed=1, fixed_length=True)
    w = RolloutWorker(env, mac, args)
    ep = w.generate_episodes(E)[0]
    if which != "rollout":
        learner = QLearner(mac, args)
        learner.train(ep, 0)
    torch.cuda.synchronize()
    v = buf.cpu().view(16, 16).numpy()
    names = SEGS[which]
    if which in ("bwd", "bwd_pipe"):
        v = v[:, 8:]
    if which == "qmix":          # forward (target mixer) in columns 0-7, loss + backward in 8-15
        show(v[:, :8], names, "qmix forward", E, args.episode_limit)
        v = v[:, 8:]
    show(v, names, which, E, args.episode_limit)


def show(v, names, which, E, T):
    print("segment shares per wave of workgroup 0 (%s, %d envs); cycles/step in the last column" % (which, E))
    print("wave " + " ".join("%9s" % n for n in names) + "   total/step")
    for wv in range(16):
        row = v[wv, :len(names)]
        tot = row.sum()
        if tot == 0:
            continue
        print("%4d " % wv + " ".join("%8.1f%%" % (100.0 * x / tot) for x in row) + "   %10.0f" % (tot / T))


if __name__ == "__main__":
    main()
