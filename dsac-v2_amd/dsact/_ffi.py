"""ctypes binding of libdsact.so (include/dsact.h). No torch types cross this boundary: device
memory is passed as integer addresses (`tensor.data_ptr()`), host arrays as numpy buffers.

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950) into
`dsac-v2_amd/lib/libdsact.so`. There is NO fallback: if it is missing, importing the update path
fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libdsact.so")
if os.environ.get("DSACT_LIB_PATH"):   # A/B runs against another build of the same ABI (scripts/): never a fallback
    LIB_PATH = os.environ["DSACT_LIB_PATH"]

MAX_HIDDEN = 6
F_SKIP_ACTOR_ON_OFF_ITERS = 1

E_NAMES = {0: "OK", -1: "E_INVALID", -2: "E_HIP", -3: "E_STATE", -4: "E_NODEVICE"}


class DsactError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * MAX_HIDDEN),
        ("batch", C.c_int32), ("global_batch", C.c_int32),
        ("auto_alpha", C.c_int32), ("delay_update", C.c_int32),
        ("gamma", C.c_double), ("tau", C.c_double), ("tau_b", C.c_double),
        ("lr_q", C.c_double), ("lr_pi", C.c_double), ("lr_alpha", C.c_double),
        ("alpha_fixed", C.c_double),
        ("min_log_std", C.c_double), ("max_log_std", C.c_double),
        ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
        ("conv_type", C.c_int32), ("img_c", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("algo", C.c_int32), ("td_bound", C.c_double),
        ("v1_unbounded", C.c_int32), ("value_act", C.c_int32), ("policy_act", C.c_int32), ("act_dist", C.c_int32),
        ("policy_std_param", C.c_int32), ("value_out_act", C.c_int32), ("policy_out_act", C.c_int32),
        ("policy_hidden", C.c_int32 * MAX_HIDDEN), ("policy_twin", C.c_int32), ("policy_n_hidden", C.c_int32),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_float), ("blocks", C.c_int32)]


# every symbol include/dsact.h declares: (name, restype, argtypes)
_P = C.c_void_p
_FP = C.POINTER(C.c_float)
_I64P = C.POINTER(C.c_int64)
SYMBOLS = [
    ("dsact_version", C.c_int, []),
    ("dsact_device_count", C.c_int, []),
    ("dsact_create", C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(_P)]),
    ("dsact_destroy", C.c_int, [_P]),
    ("dsact_last_error", C.c_char_p, [_P]),
    ("dsact_set_stream", C.c_int, [_P, _P]),
    ("dsact_get_stream", C.c_void_p, [_P]),
    ("dsact_sync", C.c_int, [_P]),
    ("dsact_online_count", C.c_size_t, [_P]),
    ("dsact_target_count", C.c_size_t, [_P]),
    ("dsact_q_count", C.c_size_t, [_P]),
    ("dsact_pi_count", C.c_size_t, [_P]),
    ("dsact_bind_arenas", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("dsact_set_action_limits", C.c_int, [_P, _FP, _FP]),
    ("dsact_get_state", C.c_int, [_P, C.POINTER(C.c_int32), _FP]),
    ("dsact_set_state", C.c_int, [_P, C.POINTER(C.c_int32), _FP]),
    ("dsact_set_hyper", C.c_int, [_P, C.c_int32, C.c_double]),
    ("dsact_buffer_create", C.c_int, [_P, C.c_int64]),
    ("dsact_buffer_add", C.c_int, [_P, C.c_int64, _FP, _FP, _FP, _FP, _FP, _FP]),
    ("dsact_buffer_size", C.c_int64, [_P]),
    ("dsact_buffer_ptr", C.c_int64, [_P]),
    ("dsact_buffer_fill_device", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    ("dsact_gather", C.c_int, [_P, _I64P, C.c_int32]),
    ("dsact_read_batch", C.c_int, [_P, _FP, _FP, _FP, _FP, _FP, _FP]),
    ("dsact_load_batch", C.c_int, [_P, _FP, _FP, _FP, _FP, _FP]),
    ("dsact_upload_index_table", C.c_int, [_P, _I64P, C.c_int32]),
    ("dsact_set_noise", C.c_int, [_P, _FP, _FP, _FP, _FP]),
    ("dsact_set_device_rng", C.c_int, [_P, C.c_uint64]),
    ("dsact_compute_grads", C.c_int, [_P, C.c_int64, C.c_uint32]),
    ("dsact_apply_update", C.c_int, [_P, C.c_int64]),
    ("dsact_step", C.c_int, [_P, C.c_int64, C.c_uint32]),
    ("dsact_graph_build", C.c_int, [_P, C.c_int32, C.c_uint32]),
    ("dsact_graph_run", C.c_int, [_P, C.c_int64, C.c_int64]),
    ("dsact_run_group", C.c_int, [_P, C.c_int64, C.c_int32, C.POINTER(C.c_int64), _FP, C.c_uint32]),
    ("dsact_dp_begin", C.c_int, [_P, C.c_int64]),
    ("dsact_dp_enqueue_grads", C.c_int, [_P, C.c_uint32]),
    ("dsact_dp_enqueue_grads_critic", C.c_int, [_P, C.c_uint32]),
    ("dsact_dp_enqueue_grads_actor", C.c_int, [_P, C.c_uint32]),
    ("dsact_dp_set_strict", C.c_int, [_P, _P]),
    ("dsact_dp_enqueue_forward", C.c_int, [_P, C.c_uint32]),
    ("dsact_dp_enqueue_backward", C.c_int, [_P, C.c_uint32]),
    ("dsact_dp_enqueue_apply", C.c_int, [_P]),
    ("dsact_comm_unique_id", C.c_int, [C.c_char_p, C.POINTER(C.c_uint8)]),
    ("dsact_comm_init", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_char_p]),
    ("dsact_comm_destroy", C.c_int, [_P]),
    ("dsact_dp_enqueue_allreduce", C.c_int, [_P]),
    ("dsact_read_stats", C.c_int, [_P, _FP]),
    ("dsact_stats_snapshot", C.c_int, [_P, C.c_int32]),
    ("dsact_stats_read", C.c_int, [_P, C.c_int32, _FP]),
    ("dsact_time_steps", C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint32, C.c_int32, _FP]),
    ("dsact_time_stage", C.c_int, [_P, C.c_int32, C.c_int32, _FP, C.POINTER(C.c_double)]),
    ("dsact_chain_active", C.c_int, [_P]),
    ("dsact_profile_step", C.c_int, [_P, C.c_int64, C.c_uint32, C.POINTER(KernelTime), C.c_int32,
                                     C.POINTER(C.c_int32)]),
    ("dsact_profile_steps", C.c_int, [_P, C.c_int64, C.c_int32, C.c_uint32, C.POINTER(KernelTime), C.c_int32,
                                      C.POINTER(C.c_int32)]),
    ("dsact_debug_read", C.c_int, [_P, C.c_char_p, _FP, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("dsact_debug_names", C.c_char_p, []),
    ("dsact_debug_set", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("dsact_debug_get", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    ("dsact_policy_forward", C.c_int, [_P, _FP, C.c_int32, _FP]),
    ("dsact_act_sample", C.c_int, [_P, _FP, _FP, _FP, _FP]),
]

_lib = None


def load():
    """Loads libdsact.so and binds every symbol; raises DsactError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DsactError(
            "libdsact.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). The DSAC-T HIP path has no CPU fallback." % LIB_PATH)
    # torch-ROCm bundles its own libamdhip64; it must be in the process BEFORE libdsact.so so that both
    # resolve to the SAME HIP runtime (device pointers and streams are shared across the boundary).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the header and the library drifted apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fptr(a):
    """numpy float32 C-contiguous array -> float* (None -> NULL)"""
    if a is None:
        return None
    assert a.dtype.name == "float32" and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_FP)
