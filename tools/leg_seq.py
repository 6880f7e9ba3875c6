#!/usr/bin/env python3
"""Several configs[] legs of bench.py in ONE process, in the given order: python tools/leg_seq.py alg:shape:envs:mixer:gemm ..."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
for spec in sys.argv[1:]:
    alg, shape, envs, md, gm = spec.split(":")
    c = bench.config_leg("leg", alg, shape, int(envs), md, gm)
    print(spec, "%.1f updates/s" % c["learner_updates_per_sec"], flush=True)
