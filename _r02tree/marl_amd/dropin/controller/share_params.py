from marl_amd.controller.share_params import SharedMAC, SeparatedMAC, SharedMACWithState, RTWMAC  # noqa: F401
