#!/usr/bin/env python3
"""One configs[] leg of bench.py in a fresh process: python tools/leg_once.py <alg> <shape> <envs> <mixer dtype> <gemm mode> [updates]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
alg, shape, envs, md, gm = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
upd = int(sys.argv[6]) if len(sys.argv) > 6 else 16
torch.cuda.set_device(0)
c = bench.config_leg("leg", alg, shape, envs, md, gm, updates=upd)
print(json.dumps({k: c[k] for k in ("workload", "learner_updates_per_sec")}), [(k["name"][:40], round(k["ms"], 3)) for k in c["kernels"]])
