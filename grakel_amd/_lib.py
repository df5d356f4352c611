"""ctypes binding of libgk_hip.so (the C ABI declared in include/gk_hip.h).

There is deliberately NO fallback: if the shared library is missing or no MI355X is
visible, every entry point raises -- a silent CPU path would void the parity claims.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgk_hip.so")


class GkError(RuntimeError):
    pass


_lib = None

_i32p = POINTER(c_int32)
_i64p = POINTER(c_int64)
_u64p = POINTER(c_uint64)
_f64p = POINTER(c_double)
_vpp = POINTER(c_void_p)

# name -> (restype, argtypes); mirrors include/gk_hip.h one to one
SIGNATURES = {
    "gk_last_error": (c_char_p, []),
    "gk_version": (c_char_p, []),
    "gk_device_count": (c_int, [POINTER(c_int)]),
    "gk_create": (c_int, [c_int, _vpp]),
    "gk_destroy": (c_int, [c_void_p]),
    "gk_set_stream": (c_int, [c_void_p, c_void_p]),
    "gk_synchronize": (c_int, [c_void_p]),
    "gk_timer_start": (c_int, [c_void_p]),
    "gk_timer_stop_ms": (c_int, [c_void_p, _f64p]),
    "gk_profile_enable": (c_int, [c_void_p, c_int]),
    "gk_profile_reset": (c_int, [c_void_p]),
    "gk_profile_get": (c_int, [c_void_p, c_char_p, _f64p, _i64p]),
    "gk_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
    "gk_get_option": (c_int, [c_void_p, c_char_p, _i64p]),
    "gk_host_alloc": (c_int, [ctypes.c_uint64, _vpp]),
    "gk_host_free": (c_int, [c_void_p]),
    "gk_batch_create": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int32, c_int, _vpp]),
    "gk_batch_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, _vpp]),
    "gk_export_state": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, _u64p]),
    "gk_import_state": (c_int, [c_void_p, c_void_p, c_uint64, _vpp]),
    "gk_batch_destroy": (c_int, [c_void_p]),
    "gk_batch_info": (c_int, [c_void_p, _i64p, _i64p, _i64p]),
    "gk_wl_relabel": (c_int, [c_void_p, c_void_p, c_int, c_int, _i64p, POINTER(c_int)]),
    "gk_wl_get_labels": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gk_wl_route": (c_int, [c_void_p, c_void_p]),
    "gk_wl_fit_transform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, _i64p, POINTER(c_int), _vpp, c_void_p]),
    "gk_wl_fitted_create": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gk_wl_fitted_destroy": (c_int, [c_void_p]),
    "gk_wl_fitted_selfk": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gk_wl_transform": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gk_wl_debug_signature": (c_int, [c_void_p, c_void_p, c_int, c_uint64, c_void_p, c_void_p]),
    "gk_features_build": (c_int, [c_void_p, c_void_p, c_int, c_int64, _vpp]),
    "gk_features_build_ex": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, _vpp]),
    "gk_features_build_range": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, _vpp]),
    "gk_features_destroy": (c_int, [c_void_p]),
    "gk_features_info": (c_int, [c_void_p, _i64p, _i64p, _i64p, _i64p, POINTER(c_int)]),
    "gk_features_operand_rows": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), _i64p, _i64p, _i64p, _i64p]),
    "gk_memcpy_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "gk_features_operand": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), _i64p]),
    "gk_features_selfk": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gk_features_debug_phi": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gk_comm_unique_id": (c_int, [c_void_p]),
    "gk_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gk_comm_destroy": (c_int, [c_void_p]),
    "gk_comm_info": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gk_batch_allgather": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int32, c_void_p, c_void_p]),
    "gk_shard_message": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                 c_void_p]),
    "gk_gram_sharded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gk_features_debug_phi_right": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "gk_gram": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "gk_gram_dev_ptr": (c_int, [c_void_p, _vpp, _i64p, _i64p]),
    "gk_gram_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "gk_gram_last_stats": (c_int, [c_void_p, _f64p, _f64p]),
    "gk_gram_block": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64]),
    "gk_gram_reset_stats": (c_int, [c_void_p]),
    "gk_block_copy": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int]),
    "gk_gram_normalize_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "gk_gram_checksum": (c_int, [c_void_p, c_void_p, _f64p, _f64p, _f64p]),
    "gk_host_copy_stats": (c_int, [c_void_p, POINTER(c_int), _f64p]),
    "gk_sp_build": (c_int, [c_void_p, c_void_p, c_void_p, c_int, _vpp, _i64p, _i64p]),
    "gk_batch_from_shards": (c_int, [c_void_p, c_int, _i64p, c_int64, c_int64, c_int64, c_void_p, c_int, _vpp]),
    "gk_core_numbers": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gk_sp_build_levels": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, _vpp, _i64p, _i64p]),
    "gk_sp_debug_apsp": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gk_sp_build_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, _vpp, _i64p, _i64p]),
    "gk_sp_debug_apsp_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
}


def load():
    """Load libgk_hip.so (built by ``__graft_entry__.build()`` / ``make -C grakel_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GkError("libgk_hip.so is not built (%s missing): run `python -c 'import "
                      "__graft_entry__ as g; g.build()'` or `make -C grakel_amd/csrc`. "
                      "There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if /opt/rocm's
    # copy gets loaded first (through this library) a later `import torch` finds "No HIP GPUs".
    # Importing torch first makes both share torch's runtime (measured on the MI355X box).  torch stays optional: without it the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().gk_last_error()
        raise GkError("libgk_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = c_int(0)
    check(load().gk_device_count(ctypes.byref(n)))
    return n.value
