"""Every process-level A/B switch of the hot path, read from the environment ONCE (at import of this module) and in one place.

They exist for measurements and variant tests; results do not depend on them beyond fp32 summation order.  The library keeps its
own table (include/marl_hip.h: marl_experiment_set / _get) - no launch path calls getenv - and the host-side switches below are
plain attributes; tests flip either through ``set()`` / the ``override()`` context manager instead of touching os.environ.

library switches (environment variable -> name, default):
    MARL_FWD_XS -> fwd_xs 1           the double-Q unroll reads the eval unroll's input-side gate sums
    MARL_FWD_DMA -> fwd_dma 0         LDS-DMA observation tile of the activation-saving unroll
    MARL_FWD_W2L -> fwd_w2l 1         six prefetch registers / fc2 fragments in LDS for wide observations
    MARL_BWD_PIPE_MAX_RT -> bwd_pipe_max_rt 4
    MARL_WGRAD_TALL -> wgrad_tall 1   LDS-staged tall weight-gradient kernel
    MARL_WIDE_RES -> wide_res 1, MARL_WIDE_RES32 -> wide_res32 0    resident-weights forward of the wide-state QMIX mixer
    MARL_UNROLL_R6 -> unroll_r6 1     non-saving split unrolls of more than 512 row tiles on csrc/agent_x6p.hip (0: csrc/agent_x6.hip everywhere)
    MARL_ROLLOUT_V1 -> rollout_v1 0   split whole-rollout kernel: 0 = by batch size, 1 = round 5 (csrc/rollout_x6_v1.hip), 2 = round 6 (csrc/rollout_x6.hip)
host switches:
    MARL_BIG_PAIR -> big_pair 0       batches beyond the pair's tile cap: eval chain and target unroll in flight together (two streams)
    MARL_NO_PAIR -> no_pair 0         the eval and target unrolls back to back instead of side by side on two streams
    MARL_NO_CHAIN -> no_chain 0       never the chain schedule (eval -> double-Q continuation beside the target unroll)
    MARL_CHAIN_SPLIT -> chain_split None    CUs of the chain side (0 = never chain, else a multiple of 8 in [8, 248])
    MARL_MLP3_KEEP -> mlp3_keep 1     the fused heads keep their hidden activations for the backward (0: recompute)
    MARL_X6_BWD_MIN_WG -> x6_bwd_min_wg 1   32-row groups from which the split BPTT kernel is used
    MARL_FORCE_REDUCER -> force_reducer 0   take the collective path with a single rank too (RCCL smoke test on a 1-GPU box)
"""
from __future__ import annotations

import contextlib
import os
import warnings

LIB_DEFAULTS = {"fwd_xs": 1, "fwd_dma": 0, "fwd_w2l": 1, "bwd_pipe_max_rt": 4, "wgrad_tall": 1, "wide_res": 1, "wide_res32": 0, "rollout_v1": 0, "unroll_r6": 1}
HOST_DEFAULTS = {"big_pair": 0, "no_pair": 0, "no_chain": 0, "chain_split": None, "mlp3_keep": 1, "x6_bwd_min_wg": 1, "force_reducer": 0}


def _env_int(name, default):
    v = os.environ.get("MARL_" + name.upper())
    if v is None or v == "":
        return default
    try:
        return int(v)
    except ValueError:
        warnings.warn("MARL_%s=%r ignored (want an integer)" % (name.upper(), v))
        return default


_lib_values = {k: _env_int(k, d) for k, d in LIB_DEFAULTS.items()}
_host_values = {k: _env_int(k, d) for k, d in HOST_DEFAULTS.items()}
if _host_values["chain_split"] is not None and not (_host_values["chain_split"] == 0 or
                                                      (8 <= _host_values["chain_split"] <= 248 and _host_values["chain_split"] % 8 == 0)):
    warnings.warn("MARL_CHAIN_SPLIT=%r ignored (want 0 or a multiple of 8 in 8..248)" % _host_values["chain_split"])
    _host_values["chain_split"] = None
_applied = None
generation = 0          # bumped by every set(): schedules captured as hipGraphs (algorithm/common.py:GraphedUpdate) are dropped when it moves


def apply(lib):
    """forward the library switches that differ from the library's defaults (called once by _lib.load())"""
    global _applied
    _applied = lib
    for k, v in _lib_values.items():
        if v != LIB_DEFAULTS[k]:
            rc = lib.marl_experiment_set(k.encode(), int(v))
            assert rc == 0, "marl_experiment_set(%s) failed" % k


def get(name):
    """host switches from this module; library switches from the LIBRARY's table once it is loaded (a C-ABI caller may have
    set them through marl_experiment_set directly - the kernels read that table, so the host decisions must too)"""
    if name in _host_values:
        return _host_values[name]
    if name in _lib_values:
        if _applied is not None:
            _lib_values[name] = int(_applied.marl_experiment_get(name.encode()))
        return _lib_values[name]
    raise KeyError(name)


def set(name, value):      # noqa: A001 - mirrors marl_experiment_set
    """set one switch for the rest of the process (tests, A/B tools); library switches go to the library at once"""
    global generation
    generation += 1
    if name in _host_values:
        _host_values[name] = value
        return
    if name not in _lib_values:
        raise KeyError(name)
    _lib_values[name] = int(value)
    from . import _lib
    rc = _lib.load().marl_experiment_set(name.encode(), int(value))
    assert rc == 0, name


@contextlib.contextmanager
def override(**kw):
    """with experiments.override(fwd_dma=1): ...   - restores the previous values on exit"""
    old = {k: get(k) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)
