from marl_amd.algorithm.q_learner import QLearner  # noqa: F401
