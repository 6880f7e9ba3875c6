"""ctypes binding of libmarl_hip.so (the C ABI declared in include/marl_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel returns an
error, this module raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C marl_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARL_HIP_LIB") or os.path.join(_HERE, "libmarl_hip.so")   # override: diagnostic builds

c_float_p = C.c_void_p   # device pointers travel as integers
c_int_p = C.c_void_p


class MarlSrc(C.Structure):
    _fields_ = [("p0", C.c_void_p), ("ld0", C.c_long), ("k0", C.c_int),
                ("p1", C.c_void_p), ("ld1", C.c_long), ("k1", C.c_int),
                ("idx", C.c_void_p), ("nhot", C.c_int), ("hot_w", C.c_int),
                ("nid", C.c_int),
                ("m0", C.c_void_p), ("ldm0", C.c_long),
                ("rpe0", C.c_long), ("bs0", C.c_long), ("off0", C.c_long),
                ("rpei", C.c_long), ("bsi", C.c_long), ("offi", C.c_long),
                ("emap0", C.c_void_p)]


class MarlGroup(C.Structure):
    _fields_ = [("groups", C.c_int), ("gs_x0", C.c_long), ("gs_x1", C.c_long), ("gs_w", C.c_long),
                ("gs_b", C.c_long), ("gs_y", C.c_long), ("gs_m0", C.c_long)]


class MarlQmixWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "w1_b", "b1", "b1_b", "w2", "w2_b", "h", "h_b", "b2_w", "b2_b")]


class MarlMlp3Weights(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3")] +
                [(n, C.c_long) for n in ("gs_w1", "gs_b1", "gs_w2", "gs_b2", "gs_w3", "gs_b3")])


class MarlQtranWeights(C.Structure):
    _fields_ = [("enc0_w", C.c_void_p), ("enc0_b", C.c_void_p), ("enc2_w", C.c_void_p), ("enc2_b", C.c_void_p),
                ("q0_w", C.c_void_p), ("q0_ld", C.c_long), ("q0_s", C.c_int),
                ("q2_w", C.c_void_p), ("q2_b", C.c_void_p), ("q4_w", C.c_void_p), ("q4_b", C.c_void_p)]


class MarlAgentGrads(C.Structure):
    _fields_ = [("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p),
                ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p)]


class MarlAgentWeights(C.Structure):
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("w_ih", C.c_void_p), ("w_hh", C.c_void_p),
                ("b_ih", C.c_void_p), ("b_hh", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
                ("H", C.c_int)]


P, I, L, F, U, SZ, D = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint, C.c_size_t, C.c_double
SRC, GRP, AW = C.POINTER(MarlSrc), C.POINTER(MarlGroup), C.POINTER(MarlAgentWeights)
AG = C.POINTER(MarlAgentGrads)
QW = C.POINTER(MarlQmixWeights)
M3 = C.POINTER(MarlMlp3Weights)
QT = C.POINTER(MarlQtranWeights)

# name -> (restype, argtypes); must list every symbol of include/marl_hip.h
SIGNATURES = {
    "marl_linear": (I, [SRC, P, L, I, P, P, L, I, I, I, I, F, GRP, P]),
    "marl_linear_wgrad": (I, [P, L, P, L, SRC, P, L, P, I, I, I, I, GRP, P, SZ, P]),
    "marl_linear_wgrad_workspace": (SZ, [I, I, I, I]),
    "marl_wgrad_slabs": (I, [I]),
    "marl_agent_unroll_fwd": (I, [AW, P, L, I, P, L, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, P, P]),
    "marl_agent_unroll_reuse_supported": (I, [I, I, I, I, I, I]),
    "marl_agent_unroll_x6_supported": (I, [I, I, I, I, I, I, I]),
    "marl_agent_unroll_x6_plain_r6": (I, [I, I, I, I, I, I, I, I]),
    "marl_agent_unroll_fwd_x6": (I, [AW, P, L, I, P, L, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, P, P]),
    "marl_agent_bwd_workspace": (SZ, [I, I, I]),
    "marl_agent_unroll_bwd_x6_supported": (I, [I, I, I, I, I]),
    "marl_agent_bwd_x6_workspace": (SZ, [I, I, I]),
    "marl_agent_unroll_bwd_x6": (I, [AW, P, P, P, P, I, P, P, P, P, AG, P, SZ, I, I, I, I, P]),
    "marl_agent_unroll_bwd": (I, [AW, P, P, P, P, P, I, P, P, P, P, P, AG, P, SZ, I, I, I, I, P]),
    "marl_replay_gather": (I, [P, I, I, I, I] + [P] * 17 + [P]),
    "marl_q_gather": (I, [P, P, P, F, P, L, I, P]),
    "marl_q_masked_max": (I, [P, P, F, P, P, L, I, P]),
    "marl_q_double_select": (I, [P, P, P, F, P, P, L, I, P]),
    "marl_q_scatter": (I, [P, P, P, P, P, L, I, I, P]),
    "marl_vec_add": (I, [P, P, P, L, P]),
    "marl_agent_sum": (I, [P, L, P, L, L, I, I, P]),
    "marl_agent_bcast": (I, [P, L, P, L, L, I, I, I, P]),
    "marl_qmix_mix_fwd": (I, [P, L, P, P, P, P, P, L, I, I, P]),
    "marl_qmix_mix_bwd": (I, [P, L, P, P, P, P, P, P, L, I, I, P]),
    "marl_qmix_tail_fwd": (I, [SRC, L, I, P, L, P, P, L, P, P, L, I, I, P]),
    "marl_qmix_fused_supported": (I, [I, I, I]),
    "marl_qmix_fused_workspace": (SZ, [L, I, I]),
    "marl_qmix_fused_fwd": (I, [QW, SRC, P, P, L, I, I, I, P]),
    "marl_qmix_fused_bwd": (I, [QW, SRC, P, P, P, QW, P, SZ, L, I, I, I, P]),
    "marl_qmix_fused_loss_bwd": (I, [QW, SRC, P, P, P, P, P, F, P, P, QW, P, P, SZ, L, I, I, I, P]),
    "marl_qmix_fused_fwd_x6": (I, [QW, SRC, P, P, L, I, I, I, P]),
    "marl_qmix_fused_bwd_x6": (I, [QW, SRC, P, P, P, QW, P, SZ, L, I, I, I, P]),
    "marl_qmix_fused_loss_bwd_x6": (I, [QW, SRC, P, P, P, P, P, F, P, P, QW, P, P, SZ, L, I, I, I, P]),
    "marl_qmix_wide_supported": (I, [I, I, I]),
    "marl_qmix_wide_workspace": (SZ, [L, I, I, I]),
    "marl_qmix_wide_fwd_kernel": (C.c_char_p, [L, I, I, I]),
    "marl_qmix_wide_fwd": (I, [QW, SRC, P, P, P, SZ, L, I, I, I, I, P]),
    "marl_qmix_wide_bwd": (I, [QW, SRC, P, P, P, QW, P, SZ, L, I, I, I, I, P]),
    "marl_qmix_wide_loss_bwd": (I, [QW, SRC, P, P, P, P, P, F, P, P, QW, P, P, SZ, L, I, I, I, I, P]),
    "marl_mlp3_supported": (I, [SRC, I, I, I, I, I]),
    "marl_mlp3_fwd": (I, [M3, SRC, P, L, L, L, I, I, I, P]),
    "marl_mlp3_bwd_workspace": (SZ, [L, I, I, I]),
    "marl_mlp3_bwd": (I, [M3, SRC, P, L, L, M3, P, SZ, L, I, I, I, P]),
    "marl_mlp3_save_floats": (SZ, [L, I, I]),
    "marl_mlp3_needs_kept": (I, [SRC, I, I]),
    "marl_mlp3_fwd_save": (I, [M3, SRC, P, L, L, P, SZ, L, I, I, I, P]),
    "marl_mlp3_bwd_saved": (I, [M3, SRC, P, L, L, M3, P, SZ, P, SZ, L, I, I, I, P]),
    "marl_mlp3_x6_supported": (I, [SRC, I, I, I, I, I]),
    "marl_mlp3_x6_fwd_save": (I, [M3, SRC, P, L, L, P, SZ, L, I, I, I, P]),
    "marl_mlp3_x6_bwd_saved": (I, [M3, SRC, P, L, L, M3, P, SZ, P, SZ, L, I, I, I, P]),
    "marl_qtran_supported": (I, [I, I, I]),
    "marl_qtran_head_fwd": (I, [QT, P, P, P, P, P, P, P, P, L, I, I, I, P]),
    "marl_qtran_head_fwd2": (I, [QT, P, P, P, P, P, P, P, P, P, P, L, I, I, I, P]),
    "marl_qtran_bwd_workspace": (SZ, [L, I]),
    "marl_qtran_state_parts_supported": (I, [I]),
    "marl_qtran_state_parts": (I, [SRC, L, I, I, P, L, P, P, P, L, P, P, P]),
    "marl_qtran_wgrad_rows_supported": (I, [I, I]),
    "marl_qtran_wgrad_rows_workspace": (SZ, [I, I]),
    "marl_qtran_wgrad_rows": (I, [SRC] + [P] * 8 + [P, L, P, P, P, P, P, P, P, SZ, L, I, I, P]),
    "marl_qtran_head_bwd": (I, [QT, P, P, P, P, P, P, P, P, P, I, P, P, P, P, SZ, L, I, I, I, P]),
    "marl_qplex_mix_fwd": (I, [P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P]),
    "marl_qplex_mix_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P]),
    "marl_first_terminated_len": (I, [P, L, I, I, P, P]),
    "marl_td_loss": (I, [P, P, P, P, P, F, P, P, P, L, P]),
    "marl_qtran_loss": (I, [P, P, P, P, P, P, P, P, P, F, F, F, P, P, P, P, P, P, L, P]),
    "marl_loss_workspace": (SZ, [L]),
    "marl_grad_sumsq": (I, [P, L, P, P, P]),
    "marl_sumsq_workspace": (SZ, [L]),
    "marl_rmsprop_step": (I, [P, P, P, L, F, F, F, F, P, P, P]),
    "marl_adam_step": (I, [P, P, P, P, L, F, F, F, F, F, F, F, P, P, P]),
    "marl_select_actions": (I, [P, P, L, P, F, U, I, P, I, P, L, I, I, I, P]),
    "marl_synth_lengths": (I, [U, I, I, P, P, I, I, P]),
    "marl_synth_observe": (I, [U, I, I, I, P, P, P, L, P, I, I, I, I, I, I, P]),
    "marl_synth_step": (I, [U, I, I, I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "marl_synth_fused_step": (I, [U, U, I, I, I, F, P, P, P, P, L, P, P, P, P, P, I, I, I, I, I, I, P]),
    "marl_synth_rollout_supported": (I, [I, I, I]),
    "marl_synth_rollout": (I, [AW, U, U, I, I, I, P, P, P, L, P, P, P, P, P, P, P, P, P, D, D, D, I, I, I, I, I, I, I, I, P]),
    "marl_synth_rollout_x6_supported": (I, [I, I, I]),
    "marl_synth_rollout_x6_plan": (I, [I, I, I, I, I, I, P]),
    "marl_synth_rollout_x6": (I, [AW, U, U, I, I, I, P, P, P, L, P, P, P, P, P, P, P, P, P, D, D, D, I, I, I, I, I, I, I, I, P]),
    "marl_hip_version": (C.c_char_p, []),
    "marl_experiment_set": (I, [C.c_char_p, I]),
    "marl_experiment_get": (I, [C.c_char_p]),
}

_lib = None


class MarlHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MarlHipError(
            "libmarl_hip.so not found at %s - the HIP hot path is mandatory (no CPU fallback). "
            "Build it with `python -c \"import __graft_entry__ as g; g.build()\"`." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    from . import experiments
    experiments.apply(lib)          # the MARL_* environment, read ONCE at import of marl_amd.experiments, goes into the library's table
    return lib


def check(code, what):
    if code != 0:
        raise MarlHipError("%s failed with hipError_t %d" % (what, code))
