"""Thin Python wrappers over the C-ABI kernels (one function per ``spn_*`` entry point).

No autograd here (see ``functional.py``) and no math: shape checks, output allocation through
PyTorch's caching allocator, pointer/stride marshalling, the current HIP stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from .lib import call, ptr, stream_ptr, require_gpu, c_int, c_long, c_float, SpnError

BF16, F32 = torch.bfloat16, torch.float32


def _dt(t_or_dtype) -> int:
    dt = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if dt == F32:
        return 0
    if dt == BF16:
        return 1
    raise SpnError(f"unsupported dtype {dt}")


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """View as 2-D [rows, cols] with unit inner stride (no copy when possible)."""
    if t.ndim != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1:
        t = t.contiguous()
    return t


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------

def gemm(a: torch.Tensor, b: torch.Tensor, *, ta: bool = False, tb: bool = False, out: Optional[torch.Tensor] = None,
         out_dtype=BF16, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         rowmask: Optional[torch.Tensor] = None, alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    """C[M,N] = residual + rowmask * (alpha * A@B + bias).

    a: [M,K] (ta=False) or stored transposed [K,M] (ta=True); b: nn.Linear-style [N,K] (tb=False) or [K,N] (tb=True).
    All operands bf16 with unit inner stride; C bf16 or fp32.
    """
    require_gpu(a, b)
    if a.dtype != BF16 or b.dtype != BF16:
        raise SpnError("gemm operands must be bf16")
    a, b = _rows2d(a), _rows2d(b)
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    if K != Kb:
        raise SpnError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    if out.shape != (M, N) or out.stride(1) != 1:
        raise SpnError("gemm: bad output tensor")
    flags = (1 if ta else 0) | (2 if tb else 0) | (4 if out.dtype == F32 else 0) | (8 if accumulate else 0)
    if residual is not None:
        residual = _rows2d(residual)
        if residual.dtype != F32 or residual.shape != (M, N):
            raise SpnError("gemm: residual must be fp32 [M,N]")
    if rowmask is not None:
        rowmask = rowmask.reshape(-1)
        if rowmask.dtype == torch.bool:
            rowmask = rowmask.view(torch.uint8)
        if rowmask.numel() != M:
            raise SpnError("gemm: rowmask must have M entries")
    if bias is not None and (bias.dtype != F32 or bias.numel() != N):
        raise SpnError("gemm: bias must be fp32 [N]")
    call("spn_gemm_bf16", ptr(a), ptr(b), ptr(out), ptr(bias), ptr(residual), ptr(rowmask), c_int(M), c_int(N), c_int(K),
         c_int(a.stride(0)), c_int(b.stride(0)), c_int(out.stride(0)), c_int(residual.stride(0) if residual is not None else 0),
         c_float(alpha), c_int(flags), c_int(1), c_long(0), c_long(0), c_long(0), stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------------------
# LayerNorm / AdaLN
# ---------------------------------------------------------------------------------------------------------

def layernorm_fwd(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                  gb: Optional[torch.Tensor] = None, out_dtype=BF16, eps: float = 1e-5, out: Optional[torch.Tensor] = None):
    """y = LN(x) * gamma + beta (affine) or gamma_t * LN(x) + beta_t with gb = [T, 2D] fp32 (adaptive)."""
    require_gpu(x)
    x2 = _rows2d(x)
    T, D = x2.shape
    y = out if out is not None else torch.empty((T, D), device=x.device, dtype=out_dtype)
    mean = torch.empty(T, device=x.device, dtype=F32)
    rstd = torch.empty(T, device=x.device, dtype=F32)
    if gb is not None:
        gb = _rows2d(gb)
        if gb.dtype != F32 or gb.shape != (T, 2 * D):
            raise SpnError("layernorm: gb must be fp32 [T, 2D]")
    call("spn_layernorm_fwd", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(gamma), ptr(beta), ptr(gb),
         c_long(gb.stride(0) if gb is not None else 0), ptr(y), c_int(_dt(y)), c_long(y.stride(0)), ptr(mean), ptr(rstd),
         c_int(T), c_int(D), c_float(eps), stream_ptr())
    return y, mean, rstd


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: Optional[torch.Tensor], gb: Optional[torch.Tensor],
                  mean: torch.Tensor, rstd: torch.Tensor, *, dres: Optional[torch.Tensor] = None, dx_dtype=F32,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None, want_dgb: bool = False):
    """Returns (dx, dgb).  dgamma/dbeta (fp32 [D]) are accumulated in place when given."""
    x2, dy2 = _rows2d(x), _rows2d(dy)
    if dy2.dtype != BF16:
        raise SpnError("layernorm_bwd: dy must be bf16")
    T, D = x2.shape
    dx = torch.empty((T, D), device=x.device, dtype=dx_dtype)
    dgb = torch.empty((T, 2 * D), device=x.device, dtype=BF16) if want_dgb else None
    if gb is not None:
        gb = _rows2d(gb)
    if dres is not None:
        dres = _rows2d(dres)
    call("spn_layernorm_bwd", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(dy2), c_long(dy2.stride(0)), ptr(gamma),
         ptr(gb), c_long(gb.stride(0) if gb is not None else 0), ptr(mean), ptr(rstd), ptr(dres),
         c_long(dres.stride(0) if dres is not None else 0), ptr(dx), c_int(_dt(dx)), c_long(dx.stride(0)), ptr(dgamma),
         ptr(dbeta), ptr(dgb), c_long(2 * D), c_int(T), c_int(D), stream_ptr())
    return dx, dgb


# ---------------------------------------------------------------------------------------------------------
# attention core
# ---------------------------------------------------------------------------------------------------------

def _bnhd_strides(t: torch.Tensor) -> Tuple[int, int, int]:
    """(batch, seq, head) element strides of a [b, n, h, 64] view with unit inner stride."""
    if t.ndim != 4 or t.shape[-1] != 64 or t.stride(-1) != 1:
        raise SpnError("attention tensors must be [b, n, h, 64] views with unit inner stride")
    return t.stride(0), t.stride(1), t.stride(2)


def _mask_u8(m: Optional[torch.Tensor]):
    if m is None:
        return None
    m = m.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m


def attn_fwd(q, k, v, *, kmask=None, slopes=None, causal=False, scale=None):
    """q [b,nq,h,64], k/v [b,nk,kvh,64] (kvh = 1 or h), kmask [b,nk] bool, slopes [h] fp32 -> o [b,nq,h,64], lse [b,h,nq]."""
    require_gpu(q, k, v)
    b, nq, h, dh = q.shape
    nk, kvh = k.shape[1], k.shape[2]
    o = torch.empty((b, nq, h, dh), device=q.device, dtype=BF16)
    lse = torch.empty((b, h, nq), device=q.device, dtype=F32)
    ks, vs = _bnhd_strides(k), _bnhd_strides(v)
    if kvh == 1:
        ks, vs = (ks[0], ks[1], 0), (vs[0], vs[1], 0)
    strides = (c_long * 12)(*_bnhd_strides(q), *ks, *vs, *_bnhd_strides(o))
    kmask = _mask_u8(kmask)
    call("spn_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(kmask), ptr(slopes), c_int(b), c_int(h), c_int(kvh),
         c_int(nq), c_int(nk), c_int(1 if causal else 0), c_float(scale if scale is not None else dh ** -0.5), strides,
         stream_ptr())
    return o, lse


def attn_bwd(q, k, v, o, d_o, lse, *, dq, dk, dv, kmask=None, slopes=None, causal=False, scale=None, want_dslope=False):
    """Writes dq/dk/dv ([b,n,h|kvh,64] bf16 views, e.g. slices of a fused dqkv buffer); returns dslope [h] fp32 or None."""
    b, nq, h, dh = q.shape
    nk, kvh = k.shape[1], k.shape[2]
    ks, vs, dks, dvs = _bnhd_strides(k), _bnhd_strides(v), _bnhd_strides(dk), _bnhd_strides(dv)
    if kvh == 1:
        ks, vs = (ks[0], ks[1], 0), (vs[0], vs[1], 0)
        dks, dvs = (dks[0], dks[1], 0), (dvs[0], dvs[1], 0)
    if _bnhd_strides(o) != _bnhd_strides(d_o):
        d_o = d_o.contiguous()
        o = o.contiguous()
    strides = (c_long * 21)(*_bnhd_strides(q), *ks, *vs, *_bnhd_strides(o), *_bnhd_strides(dq), *dks, *dvs)
    delta = torch.empty((b, h, nq), device=q.device, dtype=F32)
    dslope = torch.zeros(h, device=q.device, dtype=F32) if want_dslope else None
    kmask = _mask_u8(kmask)
    call("spn_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
         ptr(dslope), ptr(kmask), ptr(slopes), c_int(b), c_int(h), c_int(kvh), c_int(nq), c_int(nk),
         c_int(1 if causal else 0), c_float(scale if scale is not None else dh ** -0.5), strides, stream_ptr())
    return dslope
