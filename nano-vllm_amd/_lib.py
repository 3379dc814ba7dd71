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
ABI_VERSION = 7

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class Tuning(C.Structure):
    """Mirror of `nv_tuning` (include/nanovllm_hip.h): launch-plan overrides, heuristics by default."""
    _fields_ = [("decode_target_waves", _i32), ("decode_chunk", _i32), ("decode_work_item", _i32),
                ("decode_score_path", _i32), ("decode_mfma_depth", _i32), ("decode_single_launch", _i32),
                ("prefill_split", _i32), ("prefill_kernel", _i32)]
    DEFAULTS = dict(decode_target_waves=0, decode_chunk=0, decode_work_item=-1, decode_score_path=-1,
                    decode_mfma_depth=-1, decode_single_launch=-1, prefill_split=0, prefill_kernel=-1)


# name -> (restype, argtypes); must mirror include/nanovllm_hip.h exactly
PROTOTYPES = {
    "nv_strerror": (C.c_char_p, [_i32]),
    "nv_last_error": (C.c_char_p, []),
    "nv_abi_version": (_i32, []),
    "nv_set_tuning": (_i32, [C.POINTER(Tuning)]),
    "nv_get_tuning": (_i32, [C.POINTER(Tuning)]),
    "nv_store_kvcache": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i64, _i64, _vp]),
    "nv_attn_decode_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "nv_attn_decode_paged": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                    _i32, _i64, _i32, _f32, _i64, _i64, _vp, _i64, _vp]),
    "nv_attn_decode_suggest_chunk": (_i32, [_i32, _i32, _i32, _i32, _i64]),
    "nv_attn_decode_workspace_bytes_chunked": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "nv_attn_decode_paged_chunked": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                            _i32, _i64, _i32, _f32, _i64, _i64, _vp, _i64, _i32, _i32, _vp]),
    "nv_attn_decode_paged_fused_chunked": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _i64, _f32, _f32,
                                                  _i64, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp]),
    "nv_attn_decode_static_scratch_bytes": (_i64, []),
    "nv_attn_decode_paged_static": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                           _i32, _i64, _i32, _f32, _i64, _i64, _vp]),
    "nv_attn_decode_paged_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                          _i32, _i32, _i32, _i32, _i32, _i64, _i32, _i64, _f32, _f32, _i64, _i64,
                                          _i64, _i64, _vp, _i64, _vp]),
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


def set_tuning(**fields) -> None:
    """Override launch-plan heuristics process-wide (`nv_set_tuning`); no arguments restores them.
    Unknown field names raise.  A measurement / test aid: the product path never calls it."""
    if not fields:
        check(lib().nv_set_tuning(None), "nv_set_tuning")
        return
    unknown = set(fields) - set(Tuning.DEFAULTS)
    if unknown:
        raise ValueError(f"unknown tuning fields: {sorted(unknown)}")
    t = Tuning(**{**Tuning.DEFAULTS, **fields})
    check(lib().nv_set_tuning(C.byref(t)), "nv_set_tuning")


def get_tuning() -> dict:
    t = Tuning()
    check(lib().nv_get_tuning(C.byref(t)), "nv_get_tuning")
    return {name: getattr(t, name) for name, _ in Tuning._fields_}
