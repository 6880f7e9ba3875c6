from marl_amd.common.arguments import *  # noqa: F401,F403
from marl_amd.common.arguments import get_common_args, get_mixer_args  # noqa: F401
