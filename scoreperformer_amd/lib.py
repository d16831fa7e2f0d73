"""ctypes binding of the C-ABI library ``libspn.so`` (see ``include/spn.h``).

The product path has NO fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised.  PyTorch tensors only provide device memory and the
current HIP stream; every signature is plain pointers, sizes and strides.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPN_LIB") or os.path.join(_HERE, "libspn.so")   # SPN_LIB: A/B a variant build (tools/)
_lib: Optional[ctypes.CDLL] = None

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float


class SpnError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libspn.so (built by ``scoreperformer_amd.build``); raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpnError(f"{LIB_PATH} not found: run `python -m scoreperformer_amd.build` "
                           f"(the HIP extension is mandatory; there is no fallback path)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.spn_last_error.restype = ctypes.c_char_p
        _lib.spn_attn_dropbits_elems.restype = ctypes.c_long
        _lib.spn_attn_band_elems.restype = ctypes.c_long
        _lib.spn_gemm_workspace_bytes.restype = ctypes.c_size_t
        _lib.spn_tuning_name.restype = ctypes.c_char_p
        if _lib.spn_abi_version() != ABI_VERSION:
            raise SpnError(f"{LIB_PATH} has ABI version {_lib.spn_abi_version()}, this package needs {ABI_VERSION}: rebuild it")
        _apply_env_tuning(_lib)
    return _lib


ABI_VERSION = 10
TUNING_EPOCH = 0

# The library itself never reads the environment (include/spn.h): tuning knobs are set explicitly.  For A/B runs from the
# shell this binding maps SPN_<KNOB> variables onto spn_set_tuning once, at load (e.g. SPN_ATTN_BAND=0, SPN_GEMM_VARIANT=9).
_ENV_ALIASES = {"gemm_split_blocks": ("SPN_GEMM_SPLIT_BLOCKS", "SPN_GEMM_PP_SPLIT_BLOCKS")}


def _apply_env_tuning(lib) -> None:
    for i in range(lib.spn_tuning_count()):
        name = lib.spn_tuning_name(c_int(i)).decode()
        for env in _ENV_ALIASES.get(name, ("SPN_" + name.upper(),)):
            val = os.environ.get(env)
            if val is not None:
                lib.spn_set_tuning(name.encode(), ctypes.c_double(float(val)))


def set_tuning(name: str, value: float) -> None:
    """Set a process-wide tuning knob of libspn.so (names: csrc/tuning.h)."""
    lib = load()
    if lib.spn_set_tuning(name.encode(), ctypes.c_double(float(value))) != 0:
        raise SpnError(f"unknown tuning knob {name!r}")
    global TUNING_EPOCH
    TUNING_EPOCH += 1   # shape-keyed host caches that depend on a knob (ops._gemm_ws_bytes) key on it


def get_tuning(name: str) -> float:
    lib = load()
    out = ctypes.c_double(0.0)
    if lib.spn_get_tuning(name.encode(), ctypes.byref(out)) != 0:
        raise SpnError(f"unknown tuning knob {name!r}")
    return out.value


def ptr(t: Optional[torch.Tensor]):
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def stream_ptr() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args) -> None:
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        raise SpnError(f"{name} failed (rc={rc}): {lib.spn_last_error().decode()}")


def require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SpnError("scoreperformer_amd ops run only on an AMD GPU (HIP device tensors); "
                           "there is no CPU fallback in the product path")
