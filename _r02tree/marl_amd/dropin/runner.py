from marl_amd.runner import Runner  # noqa: F401
