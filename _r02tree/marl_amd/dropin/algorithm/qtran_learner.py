from marl_amd.algorithm.qtran_learner import QTRANLearner  # noqa: F401
