"""Checkpoint interop on the GPU: the state-dict files the reference ships
(model/{vdn,qplex,qtran_base}/2s3z/1_*.pkl, copied as data to tests/golden/ref_ckpt/) are loaded through the product's
own load_models() and evaluated on a seeded batch; expected outputs come from the REAL reference classes fed with the same
files (tests/golden/make_ckpt_golden.py).  Plus the learner-level full-resume round trip."""
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import seeded, learners

import parity

pytestmark = pytest.mark.gpu
B, T, LENGTHS = 3, 5, [5, 3, -1]


def _learner(alg, model_dir):
    from marl_amd.controller.share_params import SharedMAC
    from marl_amd.algorithm.q_learner import QLearner
    from marl_amd.algorithm.qtran_learner import QTRANLearner
    args = seeded.make_args("2s3z", alg, episode_limit=T)
    args.cuda = True
    args.model_dir = model_dir
    mac = SharedMAC(args)
    return args, mac, (QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args))


@pytest.mark.parametrize("alg", ["vdn", "qplex", "qtran_base"])
def test_reference_checkpoints_load_and_reproduce_reference_outputs(alg, golden_dir, tmp_path):
    fix = np.load(os.path.join(golden_dir, "ref_ckpt_outputs.npz"))
    dst = tmp_path / alg / "2s3z"
    shutil.copytree(os.path.join(golden_dir, "ref_ckpt", alg), dst)
    if alg == "vdn":      # the reference's VDN mixer has no parameters: its 495-byte file is an empty state dict
        torch.save({}, dst / "mixer_net_params.pkl")
    args, mac, learner = _learner(alg, str(tmp_path))
    learner.load_models()                                   # reference q_learner.py:198-209 file names
    batch = seeded.make_batch(args, B, seed=700, lengths=LENGTHS)
    assert abs(seeded.checksum(batch) - float(fix[alg + "/batch_checksum"])) < 1e-6
    mac.init_hidden(B)
    q, h = mac.get_current_q_values(batch, T)
    c = "ckpt:" + alg
    parity.close(c, "q_cur", q.cpu().numpy(), fix[alg + "/q_cur"])
    parity.close(c, "h_cur", h.cpu().numpy(), fix[alg + "/h_cur"])
    qc = torch.gather(q.cpu(), 3, torch.tensor(batch["u"])).squeeze(3)
    s = torch.tensor(batch["s"], dtype=torch.float32)
    uo = torch.tensor(batch["u_onehot"], dtype=torch.float32)
    if alg == "vdn":
        parity.close(c, "q_tot", learner.mixer(qc, s).cpu().numpy(), fix[alg + "/q_tot"])
    elif alg == "qplex":
        qd = q.cpu().clone(); qd[torch.tensor(batch["avail_u"]) == 0] = -9999999
        parity.close(c, "v_tot", learner.mixer(qc, s, is_v=True).cpu().numpy(), fix[alg + "/v_tot"])
        parity.close(c, "a_tot", learner.mixer(qc, s, actions=uo, max_q_i=qd.max(dim=3)[0], is_v=False).cpu().numpy(),
                     fix[alg + "/a_tot"])
    else:
        parity.close(c, "joint_q", learner.mixer(s, h, uo).cpu().numpy(), fix[alg + "/joint_q"])
        parity.close(c, "v", learner.v(s, h).cpu().numpy(), fix[alg + "/v"])


@pytest.mark.parametrize("alg,opt", [("qmix", "RMS"), ("qmix", "Adam"), ("qtran_base", "RMS")])
def test_learner_resume_state_round_trip(alg, opt, tmp_path):
    """train 3 updates, save, train 2 more; a fresh learner restored from the file repeats the last 2 bit for bit
    (parameters, both targets, optimizer statistics and step count travel; step 200 crosses a target sync)."""
    def fresh():
        args, mac, learner = _learner(alg, str(tmp_path))
        args.optimizer = opt
        return args, learner
    torch.manual_seed(5)
    args, a = fresh()
    if opt == "Adam":
        from marl_amd.algorithm.common import FusedOptimizer
        a.optimizer = FusedOptimizer(a._flat, "Adam", a.lr, args.grad_norm_clip)
    steps = [0, 1, 200, 201, 202]
    batches = [seeded.make_batch(args, 4, seed=40 + i, lengths=[5, 2, -1, 4]) for i in range(5)]
    for i in range(3):
        a.train(learners.clone_batch(batches[i]), steps[i])
    ck = str(tmp_path / "learner_resume.pt")
    a.save_resume(ck)
    tail = [a.train(learners.clone_batch(batches[i]), steps[i]) for i in (3, 4)]
    torch.manual_seed(99)                                   # different init: everything must come from the file
    _, b = fresh()
    if opt == "Adam":
        from marl_amd.algorithm.common import FusedOptimizer
        b.optimizer = FusedOptimizer(b._flat, "Adam", b.lr, args.grad_norm_clip)
    b.load_resume(ck)
    assert [b.train(learners.clone_batch(batches[i]), steps[i]) for i in (3, 4)] == tail
    assert torch.equal(a._flat.flat, b._flat.flat) and torch.equal(a.target_net.agent._flat.flat, b.target_net.agent._flat.flat)
    with pytest.raises(ValueError):
        _, other = _learner("vdn", str(tmp_path))
        other.load_resume(ck)


def test_rccl_single_rank_collectives():
    """One-node RCCL smoke (backend "nccl" IS RCCL on ROCm): a 1-rank process group on the GPU carries the learner's
    real exchange - the flat [gradients | loss numerators | sum(mask)] fp32 all-reduce, the int32 MAX all-reduce of
    max_episode_len and the start-up broadcast - so a library / dtype / IPC-mode problem shows up here and not first
    in the driver's multi-GPU run.  Run in a child process (the process group must not leak into other tests)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", MARL_FORCE_REDUCER="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from oracle import seeded, learners
from test_gpu_learners import build_product
case = ("x", "2s3z", "qmix", 4, 6, [5, 3, -1, 4], {})
args, mac, learner = build_product(case)
assert learner.reducer.enabled
losses = [learner.train(learners.clone_batch(seeded.make_batch(args, 4, seed=100 + i, lengths=[5, 3, -1, 4])), i) for i in range(2)]
dist.destroy_process_group()
from marl_amd import experiments
experiments.set("force_reducer", 0)          # (the environment is read once, at import: later changes go through experiments.set)
args2, mac2, single = build_product(case)
assert not single.reducer.enabled
ref = [single.train(learners.clone_batch(seeded.make_batch(args2, 4, seed=100 + i, lengths=[5, 3, -1, 4])), i) for i in range(2)]
assert losses == ref, (losses, ref)
print("RCCL_OK", losses)
""" % (root, root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
