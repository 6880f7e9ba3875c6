#!/usr/bin/env python3
"""Checkpoint interop fixture: load the state-dict files the reference SHIPS
(model/{vdn,qplex,qtran_base}/2s3z/1_*.pkl) into the REAL reference classes, run one seeded batch through them and
store the outputs.  Run in the build container only:
    python tests/golden/make_ckpt_golden.py
Writes tests/golden/ref_ckpt_outputs.npz and copies the checkpoint files it used (data, not source) to
tests/golden/ref_ckpt/<alg>/ so the GPU test can load the very same bytes through the product's load_models()."""
import os
import shutil
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MARL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.argv = ["x"]
np.float = float
np.long = int
sys.modules.setdefault("gym", types.SimpleNamespace(Env=object))

from oracle import seeded  # noqa: E402
from controller.share_params import SharedMAC  # noqa: E402  (reference)
from algorithm.q_learner import QLearner  # noqa: E402
from algorithm.qtran_learner import QTRANLearner  # noqa: E402

th.set_num_threads(1)
B, T = 3, 5
LENGTHS = [5, 3, -1]
FILES = {"vdn": ("rnn_net",), "qplex": ("rnn_net", "mixer_net"), "qtran_base": ("rnn_net", "mixer_net", "v_net")}


def main():
    out = {}
    for alg, kinds in FILES.items():
        args = seeded.make_args("2s3z", alg, episode_limit=T)
        args.model_dir = os.path.join(REF, "model")
        mac = SharedMAC(args)
        learner = QTRANLearner(mac, args) if alg.startswith("qtran") else QLearner(mac, args)
        src = os.path.join(REF, "model", alg, "2s3z")
        dst = os.path.join(HERE, "ref_ckpt", alg)
        os.makedirs(dst, exist_ok=True)
        for kind in kinds:
            shutil.copyfile(os.path.join(src, "1_%s_params.pkl" % kind), os.path.join(dst, "%s_params.pkl" % kind))
            os.chmod(os.path.join(dst, "%s_params.pkl" % kind), 0o644)
        # the reference's own loading calls (q_learner.py:200-209, qtran_learner.py:222-235) on the shipped files
        mac.load_models(os.path.join(src, "1_rnn_net_params.pkl"))
        if "mixer_net" in kinds:
            learner.mixer.load_state_dict(th.load(os.path.join(src, "1_mixer_net_params.pkl"), map_location="cpu"))
        if "v_net" in kinds:
            learner.v.load_state_dict(th.load(os.path.join(src, "1_v_net_params.pkl"), map_location="cpu"))
        batch = seeded.make_batch(args, B, seed=700, lengths=LENGTHS)
        out[alg + "/batch_checksum"] = np.array(seeded.checksum(batch))
        tb = {k: th.tensor(v, dtype=th.long if k == "u" else th.float32) for k, v in batch.items()}
        with th.no_grad():
            mac.init_hidden(B)
            q, h = mac.get_current_q_values(tb, T)
            out[alg + "/q_cur"], out[alg + "/h_cur"] = q.numpy(), h.numpy()
            qc = th.gather(q, 3, tb["u"]).squeeze(3)
            if alg == "vdn":
                out[alg + "/q_tot"] = learner.mixer(qc, tb["s"]).numpy()
            elif alg == "qplex":
                qd = q.clone(); qd[tb["avail_u"] == 0] = -9999999
                mx = qd.max(dim=3)[0]
                out[alg + "/v_tot"] = learner.mixer(qc, tb["s"], is_v=True).numpy()
                out[alg + "/a_tot"] = learner.mixer(qc, tb["s"], actions=tb["u_onehot"], max_q_i=mx, is_v=False).numpy()
            else:
                out[alg + "/joint_q"] = learner.mixer(tb["s"], h, tb["u_onehot"]).numpy()
                out[alg + "/v"] = learner.v(tb["s"], h).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_ckpt_outputs.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
