"""ctypes binding of libnanovllm_hip.so (the C ABI declared in include/nanovllm_hip.h).

The library is built in-tree by ``make -C nano-vllm_amd/csrc`` (or ``__graft_entry__.build()``)
into ``nano-vllm_amd/lib/``.  There is no fallback: if it cannot be loaded, ``lib()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnanovllm_hip.so")
ABI_VERSION = 2

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must mirror include/nanovllm_hip.h exactly
PROTOTYPES = {
    "nv_strerror": (C.c_char_p, [_i32]),
    "nv_last_error": (C.c_char_p, []),
    "nv_abi_version": (_i32, []),
    "nv_store_kvcache": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i64, _i64, _vp]),
    "nv_attn_decode_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "nv_attn_decode_paged": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                    _i32, _i64, _i32, _f32, _i64, _i64, _vp, _i64, _vp]),
    "nv_attn_prefill_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "nv_attn_prefill_varlen": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                      _i32, _i32, _i64, _i32, _f32, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "nv_rmsnorm": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _f32, _vp]),
    "nv_silu_and_mul": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp]),
    "nv_add_rmsnorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "nv_rope": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    "nv_qknorm_rope_store": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                    _i32, _i32, _i32, _i64, _f32, _i64, _i32, _i64, _i64, _i64, _vp]),
}

_lock = threading.Lock()
_lib = None


class NanoVllmHipError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def load(path: str | None = None) -> C.CDLL:
    """Load the shared library and attach prototypes.  Raises if it is missing or stale."""
    path = path or os.environ.get("NANOVLLM_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run "
            "`make -C nano-vllm_amd/csrc` (needs hipcc, targets gfx950). There is no CPU fallback."
        )
    dll = C.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(dll, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = dll.nv_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {got}, bindings expect {ABI_VERSION}; rebuild the library")
    return dll


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load()
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        dll = lib()
        raise NanoVllmHipError(
            f"{what}: {dll.nv_strerror(rc).decode()} ({rc}): {dll.nv_last_error().decode()}"
        )
