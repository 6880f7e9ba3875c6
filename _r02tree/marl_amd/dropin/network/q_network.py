from marl_amd.network.q_network import RNNQNet  # noqa: F401
