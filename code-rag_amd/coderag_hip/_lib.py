# -*- coding: utf-8 -*-
"""ctypes binding of libcoderag_hip.so (the C ABI of include/coderag_hip.h).

There is no CPU fallback: if the library is missing or no gfx950 device is present, the
product path raises.  The library is built in-tree by code-rag_amd/build.py.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcoderag_hip.so")

DTYPE_F32, DTYPE_BF16 = 0, 1
METRIC_L2, METRIC_IP = 0, 1
ENTRY_BYTES = 16
MAX_TOP_K = 200

# every symbol include/coderag_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "crag_version", "crag_last_error", "crag_device_count",
    "crag_index_create", "crag_index_destroy", "crag_index_size", "crag_index_dim_pad",
    "crag_index_set_id_base", "crag_index_add", "crag_index_add_device", "crag_index_attach_device",
    "crag_index_set_symbols", "crag_index_finalize", "crag_index_get_rows",
    "crag_hash_name", "crag_hash_names", "crag_symbol_hashes",
    "crag_search", "crag_search_multi", "crag_search_multi_hashed", "crag_search_device", "crag_search_multi_device",
    "crag_search_device_noalloc",
    "crag_search_local_device", "crag_merge_rerank_device",
    "crag_stats", "crag_set_path", "crag_set_profiling", "crag_get_profile", "crag_debug_dense_scores",
]

_lib = None


class CragError(RuntimeError):
    pass


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so, and libtorch_hip.so asks
    for it by the unversioned file name, which the loader does not match against an already loaded
    /opt/rocm/lib/libamdhip64.so.7.  If this library came first, torch would bring a SECOND runtime
    into the process, which then finds no devices ("No HIP GPUs are available").  Loading torch's copy
    first (when a torch wheel with one is installed) makes both sides share it: its soname is the
    versioned name this library needs.  Without torch the system runtime is used as usual."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CragError(
            f"{LIB_PATH} is missing: build it with `python code-rag_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.crag_version.restype = i32
    lib.crag_last_error.restype = C.c_char_p
    lib.crag_device_count.restype = i32
    lib.crag_index_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    lib.crag_index_destroy.argtypes = [vp]
    lib.crag_index_size.argtypes = [vp]
    lib.crag_index_size.restype = i64
    lib.crag_index_dim_pad.argtypes = [vp]
    lib.crag_index_set_id_base.argtypes = [vp, i64]
    lib.crag_index_add.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.crag_index_add_device.argtypes = [vp, vp, i64, vp, vp]
    lib.crag_index_attach_device.argtypes = [vp, vp, i64, vp, vp]
    lib.crag_index_set_symbols.argtypes = [vp, vp, vp, i64]
    lib.crag_index_finalize.argtypes = [vp]
    lib.crag_index_get_rows.argtypes = [vp, i64, i64, vp]
    lib.crag_hash_name.argtypes = [C.c_char_p, i64]
    lib.crag_hash_name.restype = C.c_uint64
    lib.crag_hash_names.argtypes = [C.c_char_p, vp, i64, vp]
    lib.crag_hash_names.restype = None
    lib.crag_symbol_hashes.argtypes = [vp, C.c_char_p, i64, vp, i32]
    lib.crag_search.argtypes = [vp, vp, i32, i32, f64, C.c_char_p, vp, vp, vp, vp, vp]
    lib.crag_search_multi.argtypes = [vp, vp, i32, i32, f64, C.c_char_p, vp, vp, vp, vp, vp]
    lib.crag_search_multi_hashed.argtypes = [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp]
    lib.crag_search_device.argtypes = [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp]
    lib.crag_search_multi_device.argtypes = [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp]
    lib.crag_search_device_noalloc.argtypes = [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp]
    lib.crag_search_local_device.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    lib.crag_merge_rerank_device.argtypes = [i32, vp, i32, i32, i32, f64, vp, vp, vp, vp, vp]
    lib.crag_stats.argtypes = [vp, vp, i32]
    lib.crag_set_path.argtypes = [vp, i32]
    lib.crag_set_profiling.argtypes = [vp, i32]
    lib.crag_get_profile.argtypes = [vp, vp, i32]
    lib.crag_debug_dense_scores.argtypes = [vp, vp, i32, i32, vp, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("crag_version", "crag_device_count"):
            fn.restype = i32
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = load().crag_last_error().decode("utf-8", "replace")
        raise CragError(f"{what or 'coderag_hip'} failed ({rc}): {msg}")
    return rc
