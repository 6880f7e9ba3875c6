"""Run one of the reference's own entry scripts UNCHANGED on the MI355X classes:

    python -m marl_amd.dropin /path/to/Skylarking-MARL/matrix_game_test.py
    MARL_N_ENVS=1024 python -m marl_amd.dropin /path/to/Skylarking-MARL/main.py --map 2s3z

Python puts a script's own directory in front of PYTHONPATH, so `python matrix_game_test.py` inside the reference
checkout would import the reference's rollout / controller / algorithm modules whatever PYTHONPATH says.  This
launcher puts `marl_amd/dropin` (one-line re-exports under the reference's module paths) FIRST on sys.path and then
executes the script with runpy (which does not add the script's directory), so every hot-path import of
runner.py:3-11, matrix_game_test.py:3-10 and main.py:1-5 resolves to marl_amd; modules the drop-in does not provide
(the reference's own runner.py, its plotting helpers) are still found in the script's directory, which is appended
LAST."""
import importlib.util
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install(script_dir=None):
    """Put the drop-in module paths first on sys.path (idempotent).  `smac` falls back to the synthetic shim only
    when the real package is not importable."""
    root = os.path.dirname(os.path.dirname(HERE))
    for p in (root, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    if importlib.util.find_spec("smac") is None:
        shim = os.path.join(HERE, "smac_shim")
        if shim not in sys.path:
            sys.path.insert(1, shim)
    else:
        vectorise_real_smac(int(os.environ.get("MARL_N_ENVS", "1")))
    if script_dir and script_dir not in sys.path:
        sys.path.append(script_dir)


def vectorise_real_smac(n_envs, n_threads=None):
    """A REAL `smac` is importable: `StarCraft2Env(**kw)` of reference main.py:16-20 then builds ``n_envs`` of them
    behind marl_amd.env.host_vector.HostVectorEnv (lock-step rollout, one H2D + one D2H per step) when MARL_N_ENVS > 1;
    with MARL_N_ENVS <= 1 the script gets the plain environment and the reference's serial loop."""
    if n_envs <= 1:
        return False
    import smac.env as smac_env
    real = smac_env.StarCraft2Env
    if getattr(real, "_marl_vectorised", False):
        return True

    def StarCraft2Env(*a, **kw):
        from marl_amd.env.host_vector import HostVectorEnv
        threads = n_threads if n_threads is not None else int(os.environ.get("MARL_ENV_THREADS", str(min(n_envs, 16))))
        return HostVectorEnv((lambda: real(*a, **kw), n_envs), seed=kw.get("seed", 1) or 1, n_threads=threads)

    StarCraft2Env._marl_vectorised = True
    StarCraft2Env._marl_real = real
    smac_env.StarCraft2Env = StarCraft2Env
    return True


def main(argv):
    if not argv:
        raise SystemExit("usage: python -m marl_amd.dropin <reference script.py> [its arguments]")
    script = os.path.abspath(argv[0])
    if sys.path and sys.path[0] in ("", os.getcwd()):
        sys.path.pop(0)             # `python -m` put the working directory first: it may be the reference checkout
    install(os.path.dirname(script))
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv[1:])
