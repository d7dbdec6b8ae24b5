"""ctypes binding of the C ABI in include/ba_hip.h (libba_hip.so).

The library is the product path: there is NO CPU fallback.  If the shared
object is missing or no GPU is present, creation of a handle fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BA_HIP_LIB: developer override (a library built with in-kernel time stamps)
LIB_PATH = os.environ.get("BA_HIP_LIB") or os.path.join(_HERE, "libba_hip.so")


class BaOptions(C.Structure):
    """ba_options — mirrors reference core/solver_option_and_summary.h:47-71."""
    _fields_ = [
        ("threshold_step_size", C.c_float),
        ("threshold_cost_change", C.c_float),
        ("threshold_huber_loss", C.c_float),
        ("threshold_outlier_rejection", C.c_float),
        ("max_num_iterations", C.c_int),
        ("initial_lambda", C.c_float),
        ("decrease_ratio_lambda", C.c_float),
        ("increase_ratio_lambda", C.c_float),
        ("gauss_newton", C.c_int),
    ]


def make_options(max_iter=50, thr_step=1e-5, thr_cost=1e-5, huber=1.0,
                 outlier=2.0, lambda0=100.0, dec=0.33, inc=3.0,
                 gauss_newton=False):
    """ba_options with the defaults of reference test/test_ba.cpp:279-290
    (lambda0 100, ratios 0.33 / 3.0, Huber 1.0)."""
    o = BaOptions()
    o.gauss_newton = 1 if gauss_newton else 0
    o.threshold_step_size = thr_step
    o.threshold_cost_change = thr_cost
    o.threshold_huber_loss = huber
    o.threshold_outlier_rejection = outlier
    o.max_num_iterations = max_iter
    o.initial_lambda = lambda0
    o.decrease_ratio_lambda = dec
    o.increase_ratio_lambda = inc
    return o


class BaIterInfo(C.Structure):
    """ba_iter_info — OptimizationInfo (reference
    core/solver_option_and_summary.h:37-46) + trust-region internals."""
    _fields_ = [
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("average_reprojection_error", C.c_double),
        ("abs_gradient", C.c_double),
        ("abs_step", C.c_double),
        ("damping_term", C.c_double),
        ("iter_time_ms", C.c_double),
        ("iteration_status", C.c_int),
        ("pad_", C.c_int),
        ("rho", C.c_double),
        ("model_change", C.c_double),
        ("trial_cost", C.c_double),
    ]


class BaPoIter(C.Structure):
    _fields_ = [("cost", C.c_float), ("cost_change", C.c_float),
                ("abs_step", C.c_float)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                           C.c_int64, C.c_void_p)

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_F = C.POINTER(C.c_float)
_I32 = C.POINTER(C.c_int32)
_U8 = C.POINTER(C.c_uint8)

# name -> (restype, argtypes): every symbol declared in include/ba_hip.h
SIGNATURES = {
    "ba_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "ba_destroy": (None, [_P]),
    "ba_last_error": (C.c_char_p, []),
    "ba_set_stream": (C.c_int, [_P, _P]),
    "ba_set_cameras": (C.c_int, [_P, C.c_int, _D, _D]),
    "ba_set_poses": (C.c_int, [_P, C.c_int, _D, _U8]),
    "ba_set_points": (C.c_int, [_P, C.c_int, _D, _U8]),
    "ba_set_observations": (C.c_int, [_P, C.c_int64, _I32, _I32, _I32, _D]),
    "ba_set_shard": (C.c_int, [_P, C.c_int, C.c_int]),
    "ba_finalize": (C.c_int, [_P]),
    "ba_update_values": (C.c_int, [_P, _D, _D]),
    "ba_partition_points": (C.c_int, [C.c_int, _U8, C.c_int, _U8, C.c_int64,
                                      _I32, _I32, C.c_int, _I32]),
    "ba_set_allreduce": (C.c_int, [_P, ALLREDUCE_FN, _P]),
    "ba_reduce_buffer_size": (C.c_int64, [_P, C.c_int]),
    "ba_bind_reduce_buffer": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "ba_gather_points": (C.c_int, [_P]),
    "ba_rccl_available": (C.c_int, []),
    "ba_rccl_get_unique_id": (C.c_int, [_U8]),
    "ba_rccl_comm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, _U8, C.c_int]),
    "ba_rccl_comm_size": (C.c_int, [_P]),
    "ba_rccl_comm_calls": (C.c_int64, [_P]),
    "ba_rccl_comm_destroy": (None, [_P]),
    "ba_rccl_allreduce_hook": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
    "ba_stream_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int64]),
    "ba_stream_destroy": (None, [_P]),
    "ba_stream_set_cameras": (C.c_int, [_P, C.c_int, _D, _D]),
    "ba_stream_set_poses": (C.c_int, [_P, C.c_int, _D, _U8]),
    "ba_stream_set_points": (C.c_int, [_P, C.c_int, _D, _U8]),
    "ba_stream_set_observations": (C.c_int, [_P, C.c_int64, _I32, _I32, _I32, _D]),
    "ba_stream_finalize": (C.c_int, [_P]),
    "ba_stream_solve": (C.c_int, [_P, C.POINTER(BaOptions), C.POINTER(BaIterInfo),
                                  C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ba_stream_lm_begin": (C.c_int, [_P, C.POINTER(BaOptions)]),
    "ba_stream_lm_iterate": (C.c_int, [_P, C.c_int]),
    "ba_stream_lm_sync": (C.c_int, [_P, C.POINTER(BaIterInfo), C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ba_stream_get_poses": (C.c_int, [_P, _D]),
    "ba_stream_get_points": (C.c_int, [_P, _D]),
    "ba_stream_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_solve": (C.c_int, [_P, C.POINTER(BaOptions), C.POINTER(BaIterInfo),
                           C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ba_lm_begin": (C.c_int, [_P, C.POINTER(BaOptions)]),
    "ba_lm_iterate": (C.c_int, [_P, C.c_int]),
    "ba_lm_sync": (C.c_int, [_P, C.POINTER(BaIterInfo), C.c_int,
                             C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ba_stage_cost": (C.c_int, [_P, _D]),
    "ba_stage_linearize": (C.c_int, [_P, C.c_double, C.c_double]),
    "ba_stage_schur": (C.c_int, [_P]),
    "ba_stage_solve_reduced": (C.c_int, [_P]),
    "ba_stage_backsub_update": (C.c_int, [_P]),
    "ba_stage_scalars": (C.c_int, [_P, _D, _D, _D, _D]),
    "ba_stage_commit": (C.c_int, [_P, C.c_int]),
    "ba_enable_stage_timing": (C.c_int, [_P, C.c_int]),
    "ba_get_stage_ms": (C.c_int, [_P, _D, C.c_int]),
    "ba_num_opt_poses": (C.c_int, [_P]),
    "ba_num_opt_points": (C.c_int, [_P]),
    "ba_num_pairs": (C.c_int64, [_P]),
    "ba_num_schur_blocks": (C.c_int64, [_P]),
    "ba_num_schur_triples": (C.c_int64, [_P]),
    "ba_get_poses": (C.c_int, [_P, _D]),
    "ba_get_points": (C.c_int, [_P, _D, _U8]),
    "ba_get_A": (C.c_int, [_P, _D, _D]),
    "ba_get_C": (C.c_int, [_P, _D, _D]),
    "ba_get_Cinv": (C.c_int, [_P, _D, _D]),
    "ba_get_pairs": (C.c_int, [_P, _I32, _I32, _D]),
    "ba_get_S": (C.c_int, [_P, _D, _D]),
    "ba_get_xy": (C.c_int, [_P, _D, _D]),
    "ba_kernel_count": (C.c_int, []),
    "ba_kernel_name": (C.c_char_p, [C.c_int]),
    "ba_get_kernel_ms": (C.c_int, [_P, _D, C.POINTER(C.c_int64), C.c_int]),
    "ba_get_dense_info": (C.c_int, [_P, _D]),
    "ba_get_schur_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_get_lin_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_get_mask_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_get_dropped_pivots": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int]),
    "ba_dense_spd_solve": (C.c_int, [_P, C.c_int, _D, _D, _D, _D]),
    "ba_pose_only_mono6": (C.c_int, [_P, _F, _F, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_float, _F, _U8,
                                     C.POINTER(BaOptions), C.POINTER(BaPoIter),
                                     C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), _F]),
    "ba_pose_only_stereo6": (C.c_int, [_P, _F, _F, _F, C.c_int, _F, _F, _F, _F,
                                       _U8, _U8, C.POINTER(BaOptions),
                                       C.POINTER(BaPoIter), C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       _F]),
}

_lib = None


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP/HSA runtime.  The PyTorch-ROCm wheel
    bundles its own libamdhip64.so (SONAME libamdhip64.so.7, requested by torch
    as "libamdhip64.so"), while libba_hip.so requests "libamdhip64.so.7": if
    the system copy were loaded first, a later `import torch` would bring in a
    second runtime and find no GPU.  When torch is installed (bench.py and the
    multi-GPU path use it for streams and torch.distributed) its copy is
    therefore loaded first, without importing torch; both libraries then share
    it.  Without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libba_hip.so and attach the C-ABI signatures (raises if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libba_hip.so is not built (%s). Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C "
            "bundle_adjustment_solver_amd/csrc`. The HIP path has no CPU "
            "fallback." % LIB_PATH)
    _preload_shared_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class BaError(RuntimeError):
    pass


def check(rc, what=""):
    if rc < 0:
        msg = load().ba_last_error()
        raise BaError("%s failed: %s" % (what, msg.decode() if msg else "?"))
    return rc
