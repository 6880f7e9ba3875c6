from marl_amd.utils.logging import Logger, get_logger  # noqa: F401
