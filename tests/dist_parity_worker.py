"""Worker for tests/test_gpu_dist.py: each rank trains on its shard of a seeded global batch.
Launched by torch.distributed.run; rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    alg, shape = sys.argv[1], sys.argv[2]
    gemm_mode = sys.argv[3] if len(sys.argv) > 3 else None
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("MARL_BENCH_BACKEND", "gloo")
    if backend == "nccl":          # RCCL: one GPU per rank
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
    else:                          # gloo moving CUDA tensors: every rank on GPU 0 (1-GPU box)
        torch.cuda.set_device(0)
        dist.init_process_group(backend=backend)
    from oracle import seeded
    from test_gpu_learners import build_product
    B, T = 6, 6
    lengths = [6, 2, 3, 4, 2, 3]
    case = ("x", shape, alg, B, T, lengths, {})
    args, mac, learner = build_product(case, gemm_mode)
    assert learner.reducer.enabled
    losses = []
    per = B // world
    for i in range(3):
        full = seeded.make_batch(args, B, seed=100 + i, lengths=lengths)
        shard = {k: v[rank * per:(rank + 1) * per].copy() for k, v in full.items()}
        losses.append(learner.train(shard, i))
    flat = learner._flat.flat.double().cpu().numpy()
    if rank == 0:
        print(json.dumps({"losses": losses, "T": learner.max_episode_len,
                          "param_sum": float(flat.sum()), "param_abs": float(np.abs(flat).sum())}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
