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
    return _lib


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
