import sys, time, ctypes, torch
sys.path.insert(0,'.')
import bench
from marl_amd import _lib
from marl_amd.controller.share_params import SharedMAC
from marl_amd.rollout import RolloutWorker
from marl_amd.env.synthetic_smac import SyntheticSMACEnv
lib=_lib.load()
args=bench.make_args('qmix','2s3z',0)
mac=SharedMAC(args); mac.cuda()
env=SyntheticSMACEnv(4096,5,80,120,11,120,seed=1,fixed_length=True)
w=RolloutWorker(env,mac,args)
for rt in (1,2,3,5):
    lib.marl_debug_set_rt_single(rt)
    w.generate_episodes(4096); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(3): w.generate_episodes(4096)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/3
    print("rt",rt,"ms/rollout %.2f"%(dt*1e3),"env-steps/s %.1fM"%(4096*120/dt/1e6))
