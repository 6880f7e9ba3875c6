"""marl_amd - MI355X-native hot path for Skylarking/MARL (rollout + GRU agent unroll + mixers).

Host code mirrors the reference's Python class surface (SharedMAC / QLearner / QTRANLearner /
ReplayBuffer / RolloutWorker); all compute is hand-written HIP in ``marl_amd/csrc`` behind the
C ABI of ``include/marl_hip.h``.  There is no CPU fallback.
"""
__version__ = "0.1.0"
