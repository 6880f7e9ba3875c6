from marl_amd.rollout import *  # noqa: F401,F403
from marl_amd.rollout import RolloutWorker, EpisodeBatch  # noqa: F401
