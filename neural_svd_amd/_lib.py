"""ctypes binding of ``libnsvd_hip.so`` (C ABI declared in ``include/nsvd.h``).

There is deliberately NO fallback: if the shared library is missing or does not export the
expected ABI, importing the ops raises, and every op raises when handed a non-GPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

NSVD_MAX_LAYERS = 8
ABI_VERSION = 2

EINVAL = -10001
EUNSUPPORTED = -10002

POT_HYDROGEN, POT_HARMONIC = 0, 1
MASK_CUSTOM, MASK_SEQUENTIAL, MASK_JOINT = 0, 1, 2
PATH_AUTO, PATH_GENERIC, PATH_FUSED, PATH_FUSED_BF16X3 = 0, 1, 2, 3
NORMALIZE_L2_BALL, NORMALIZE_L2_SPHERE = 0, 1
FEATURES_READY = 0x100
W_PLANES_READY = 0x200

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# NSVD_LIB_PATH: diagnostic builds only (e.g. the stamped kernels of scripts/dev/stamps.py)
LIB_PATH = os.environ.get("NSVD_LIB_PATH") or os.path.join(_PKG_DIR, "libnsvd_hip.so")


class ModelDesc(C.Structure):
    _fields_ = [("L", C.c_int32), ("D", C.c_int32), ("m", C.c_int32), ("nlayers", C.c_int32),
                ("dims", C.c_int32 * NSVD_MAX_LAYERS), ("has_exp_mask", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("fourier_B", C.c_void_p), ("W", C.c_void_p * NSVD_MAX_LAYERS),
                ("b", C.c_void_p * NSVD_MAX_LAYERS), ("scales", C.c_void_p)]


class Problem(C.Structure):
    _fields_ = [("potential", C.c_int32), ("charge_or_k", C.c_float), ("scale_kinetic", C.c_float),
                ("eps", C.c_float), ("op_scale", C.c_float), ("op_shift", C.c_float), ("sigma", C.c_float),
                ("hard_mul_const", C.c_float), ("use_importance", C.c_int32)]


class TowerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "g1", "be1", "rm1", "rv1", "W2", "b2", "g2", "be2", "rm2", "rv2")]


class CdkStepDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("d0", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("slope", C.c_float),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("mu", C.c_float), ("normalize_mode", C.c_int32),
                ("set_first_mode_const", C.c_int32), ("lr", C.c_double), ("momentum", C.c_double),
                ("max_grad_norm", C.c_double), ("first_step", C.c_int32), ("gemm_bf16", C.c_int32),
                ("reserved0", C.c_int32), ("grad_scaler", C.c_void_p)]


class GradScalerState(C.Structure):
    """host mirror of the DEVICE-resident nsvd_grad_scaler (include/nsvd.h)"""
    _fields_ = [("scale", C.c_float), ("growth_factor", C.c_float), ("backoff_factor", C.c_float),
                ("growth_interval", C.c_int32), ("growth_tracker", C.c_int32), ("steps_ok", C.c_int32),
                ("steps_skipped", C.c_int32), ("last_found_inf", C.c_int32)]


class Rmsprop(C.Structure):
    _fields_ = [("sq", Params), ("ema", Params), ("lr", C.c_double), ("alpha", C.c_double), ("eps", C.c_double),
                ("ema_decay", C.c_double), ("has_ema", C.c_int32), ("state", C.c_void_p)]


class StepStateCur(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "alpha", "one_minus_alpha", "eps", "one_minus_decay", "grad_scale")]


class StepState(C.Structure):
    """host mirror of the DEVICE-resident nsvd_step_state (sizes / offsets only: the library fills and advances it)"""
    _fields_ = [("step", C.c_uint64), ("T_max", C.c_uint64), ("lr0", C.c_double), ("eta_min", C.c_double),
                ("alpha", C.c_double), ("eps", C.c_double), ("ema_decay", C.c_double), ("cur", StepStateCur),
                ("reserved", C.c_uint64)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/nsvd.h
_P, _I, _F, _Z, _Dbl = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_double
SIGNATURES = {
    "nsvd_abi_version": (_I, []),
    "nsvd_path_name": (C.c_char_p, [C.POINTER(ModelDesc), _I, _I]),
    "nsvd_path_name_for": (C.c_char_p, [C.POINTER(ModelDesc), C.POINTER(Problem), _I, _I]),
    "nsvd_workspace_bytes": (_Z, [C.POINTER(ModelDesc), _I]),
    "nsvd_fourier_features": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _I, _P]),
    "nsvd_operator_forward": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P, _P,
                                   _P, _Z, _I, _I, _P]),
    "nsvd_operator_features": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P, _Z, _I,
                                    _I, _P]),
    "nsvd_operator_sample_features": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), C.c_uint64,
                                           C.c_uint64, _P, _I, _P, _Z, _I, _I, _P]),
    "nsvd_operator_backward": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P,
                                    C.POINTER(Params), _P, _Z, _I, _P]),
    "nsvd_model_forward": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), _P, _I, _F, _P, _P, _Z, _I, _P]),
    "nsvd_model_backward": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), _P, _I, _P, C.POINTER(Params), _P, _Z, _P]),
    "nsvd_model_backward_evd_step": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), _P, _I, _P, _P, _I, _P, _P, _P, _I, _P,
                                          _I, _I, _F, _P, C.POINTER(Params), C.POINTER(Rmsprop), _P, _Z, _P]),
    "nsvd_evd_scratch_bytes": (_Z, [_I, _I]),
    "nsvd_evd_moments": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "nsvd_evd_loss_grad": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P]),
    "nsvd_evd_partial": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "nsvd_evd_gather_heads": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nsvd_evd_gather_head_blocks": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nsvd_operator_backward_evd": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P, _P,
                                        _I, _P, _P, _P, _I, _P, _I, _I, _F, _P, C.POINTER(Params), _P, _Z, _I, _P]),
    "nsvd_operator_backward_evd_heads": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P,
                                              _P, _I, _P, _P, _P, _I, _P, _I, _I, _F, _P, C.POINTER(Params), _P, _Z,
                                              _I, _I, _I, _P]),
    "nsvd_backward_head_window_ok": (_I, [C.POINTER(ModelDesc), C.POINTER(Problem), _I, _I, _I]),
    "nsvd_operator_backward_evd_step": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I, _P,
                                             _P, _I, _P, _P, _P, _I, _P, _I, _I, _F, _P, C.POINTER(Params),
                                             C.POINTER(Rmsprop), _P, _Z, _I, _P]),
    "nsvd_operator_backward_evd_step_window": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I,
                                                    _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _F, _P, C.POINTER(Params),
                                                    C.POINTER(Rmsprop), _P, _Z, _I, _I, _I, _I, _P, C.c_ulonglong,
                                                    C.c_ulonglong, _P, _P, _Z, _P]),
    "nsvd_operator_backward_evd_step_next": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), _P, _I,
                                                  _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _F, _P, C.POINTER(Params),
                                                  C.POINTER(Rmsprop), _P, _Z, _I, C.c_uint64, C.c_uint64, _P, _P, _Z,
                                                  _P]),
    "nsvd_evd_loss_fused": (_I, [_P, _P, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P]),
    "nsvd_rmsprop_ema_step": (_I, [_P, _P, _P, _P, _Z, _Dbl, _Dbl, _Dbl, _Dbl, _Dbl, _P]),
    "nsvd_step_state_init": (_I, [_P, _Dbl, _Dbl, C.c_uint64, _Dbl, _Dbl, _Dbl, C.c_uint64, _P]),
    "nsvd_step_state_begin": (_I, [_P, _P]),
    "nsvd_rmsprop_ema_step_dev": (_I, [_P, _P, _P, _P, _Z, _P, _Dbl, _I, _P]),
    "nsvd_operator_sample_features_dev": (_I, [C.POINTER(ModelDesc), C.POINTER(Params), C.POINTER(Problem), C.c_uint64,
                                               C.c_uint64, _P, _P, _I, _P, _Z, _I, _I, _P]),
    "nsvd_profile_next_forward": (_I, [_P, _P]),
    "nsvd_model_workspace_bytes": (_Z, [C.POINTER(ModelDesc), _I]),
    "nsvd_kernel_apply_workspace_bytes": (_Z, [_I, _I, _I]),
    "nsvd_kernel_apply": (_I, [_P, _Z, _I, _P, _I, _P, _I, _P, _I, _F, _P, _P, _Z, _P]),
    "nsvd_cdk_workspace_bytes": (_Z, [_I, _I, _I]),
    "nsvd_cdk_loss_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "nsvd_cdk_loss_backward": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "nsvd_spectrum_accumulate": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _F, _P, _P, _P]),
    "nsvd_spectrum_accumulate_f64": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _F, _P, _P, _P]),
    "nsvd_spectrum_accumulate_const_f64": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _F, _P, _P, _P]),
    "nsvd_step_emits_planes": (_I, [C.POINTER(ModelDesc), _I, _I]),
    "nsvd_gemm_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, C.c_long, C.c_long, C.c_long, _I, _I, _I, _I, C.c_long, _P, _P]),
    "nsvd_to_bf16": (_I, [_P, _P, _Z, _P]),
    "nsvd_row_normalize_forward": (_I, [_P, _I, _I, _F, _I, _P, _P]),
    "nsvd_row_normalize_backward": (_I, [_P, _P, _I, _I, _F, _I, _P, _P]),
    "nsvd_tower_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "nsvd_tower_mixed_supported": (_I, [_I, _I, _I, _I]),
    "nsvd_tower_mixed_fused": (_I, [_I, _I, _I, _I, _F]),
    "nsvd_grad_scaler_init": (_I, [_P, _F, _F, _F, _I, _P]),
    "nsvd_tower_forward": (_I, [_P, C.POINTER(TowerParams), _I, _I, _I, _I, _F, _F, _F, _I, _I, _P, _P, _Z, _P]),
    "nsvd_tower_forward_phase": (_I, [_P, C.POINTER(TowerParams), _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _P, _P, _Z, _P]),
    "nsvd_tower_y2_offset": (_Z, [_I, _I, _I, _I]),
    "nsvd_cdk_step_workspace_bytes": (_Z, [C.POINTER(CdkStepDesc)]),
    "nsvd_cdk_step": (_I, [C.POINTER(CdkStepDesc), _P, _P, C.POINTER(TowerParams), C.POINTER(TowerParams), _P, _P, _P,
                           _P, _P, _P, _Z, _P]),
    "nsvd_tower_backward": (_I, [_P, C.POINTER(TowerParams), _P, _I, _I, _I, _I, _F, _I, C.POINTER(TowerParams), _P,
                                 _Z, _P]),
}

_lib: Optional[C.CDLL] = None


class NsvdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the shared library. Raises NsvdError when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: libnsvd_hip.so needs libamdhip64.so.7 by name; with torch loaded that name resolves to the HIP runtime
    # torch itself runs on (its bundled copy) - ONE runtime per process. Loaded the other way round, the library pulls
    # /opt/rocm's runtime in, torch brings its own, and kernels launched through one cannot see memory allocated
    # through the other (HIP error 100 at the first launch: found by build() + smoke() in one process)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NsvdError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C neural_svd_amd/csrc`). "
            f"neural_svd_amd has no CPU / eager fallback by design.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NsvdError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.nsvd_abi_version()
    if v != ABI_VERSION:
        raise NsvdError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}; rebuild the extension")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == EINVAL:
        raise NsvdError(f"{what}: invalid argument (NSVD_EINVAL)")
    if rc == EUNSUPPORTED:
        raise NsvdError(f"{what}: unsupported configuration (NSVD_EUNSUPPORTED)")
    raise NsvdError(f"{what}: HIP error {-rc}")
