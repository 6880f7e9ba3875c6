"""Case table shared by the golden generator and the tests (kept identical to
tests/golden/make_golden.py:CASES)."""
import os

import numpy as np

from oracle import seeded, learners

CASES = [
    ("vdn_matrix", "matrix", "vdn", 32, 1, [1] * 32, {}),
    ("qmix_2s3z", "2s3z", "qmix", 4, 6, [5, 3, -1, 4], {}),
    ("qmix_2s3z_adam", "2s3z", "qmix", 3, 5, [5, 2, 4], {"optimizer": "Adam"}),
    ("qmix_2s3z_hyper2", "2s3z", "qmix", 3, 5, [-1, -1, -1], {"two_hyper_layers": True}),
    ("vdn_2s3z_nodq", "2s3z", "vdn", 3, 5, [3, 5, 4], {"double_q": False}),
    ("qplex_2s3z", "2s3z", "qplex", 4, 6, [6, 3, -1, 4], {}),
    ("qplex_2s3z_nodq", "2s3z", "qplex", 3, 4, [4, 2, 3], {"double_q": False}),
    ("qtran_3s5z", "3s5z", "qtran_base", 4, 6, [6, 2, -1, 5], {}),
    ("qmix_MMM2", "MMM2", "qmix", 3, 5, [5, 3, 4], {}),
    # round 3: the shapes that left the marl_linear composition (heads with 322 / 502 input columns, 320-wide hypernet heads)
    ("qplex_MMM2", "MMM2", "qplex", 3, 4, [4, 2, 3], {}),
    ("qmix_MMM2_hyper2", "MMM2", "qmix", 3, 5, [5, 2, 4], {"two_hyper_layers": True}),
    ("qplex_3s5z", "3s5z", "qplex", 3, 4, [3, -1, 4], {}),
]
TRAIN_STEPS = [0, 1, 200, 201]


def load_fixture(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def case_states(case):
    """Seeded numpy weights for a case: (args, agent, mixer, v, extra)."""
    name, shape, alg, B, T, lengths, over = case
    args = seeded.make_args(shape, alg, episode_limit=T, **over)
    agent = seeded.seeded_state(seeded.agent_param_shapes(args), seed=11)
    mshapes = seeded.mixer_param_shapes(args)
    mixer = seeded.seeded_state(mshapes, seed=12) if mshapes else {}
    v = extra = None
    if alg.startswith("qtran"):
        v = seeded.seeded_state(seeded.qtran_v_param_shapes(args), seed=13)
        extra = seeded.seeded_state(seeded.qmix_param_shapes(args), seed=14)
    return args, agent, mixer, v, extra


def build_oracle_state(case):
    args, agent, mixer, v, extra = case_states(case)
    return args, learners.LearnerState(args, agent, mixer, v, extra)
