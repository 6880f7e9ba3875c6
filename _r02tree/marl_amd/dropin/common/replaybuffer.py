from marl_amd.common.replaybuffer import ReplayBuffer  # noqa: F401
