"""`from smac.env import StarCraft2Env` (reference main.py:2,16-20) -> the synthetic batched env with the map's dims.
The number of lock-step environments comes from MARL_N_ENVS (default 32)."""
import os

from marl_amd.env.synthetic_smac import SyntheticSMACEnv
from marl_amd.main import MAPS


def StarCraft2Env(map_name="2s3z", step_mul=8, difficulty="7", game_version="latest", replay_dir="", seed=1, **_):
    n, o, s, a, t = MAPS[map_name]
    return SyntheticSMACEnv(int(os.environ.get("MARL_N_ENVS", "32")), n, o, s, a, t, seed=seed)
