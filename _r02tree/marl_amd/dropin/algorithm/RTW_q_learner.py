"""Name kept so that `from algorithm.RTW_q_learner import RTWQLearner` (reference runner.py:9) resolves;
the RTW research variant is outside the hot path."""


class RTWQLearner:
    def __init__(self, *a, **k):
        raise NotImplementedError("RTWQLearner is out of scope; run with --RTW ''")
