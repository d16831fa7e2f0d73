"""Thin Python wrappers over the C-ABI kernels (one function per ``spn_*`` entry point).

No autograd here (see ``functional.py``) and no math: shape checks, output allocation through
PyTorch's caching allocator, pointer/stride marshalling, the current HIP stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import lib as _lib_mod
from .lib import load,  call, ptr, stream_ptr, require_gpu, c_int, c_long, c_float, SpnError

BF16, F32 = torch.bfloat16, torch.float32


def _dt(t_or_dtype) -> int:
    dt = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if dt == F32:
        return 0
    if dt == BF16:
        return 1
    raise SpnError(f"unsupported dtype {dt}")


def _fit8(t: torch.Tensor, k_rows: bool, K8: int) -> torch.Tensor:
    """GEMM operand with 8-element aligned rows, a 16-byte aligned base and K zero-padded to K8 (a copy only when needed).
    k_rows: K is the row dimension (the operand is stored transposed)."""
    rows, cols = t.shape
    ok = t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 and (rows == K8 if k_rows else cols == K8)
    if ok:
        return t
    R, C = (K8, (cols + 7) // 8 * 8) if k_rows else (rows, K8)
    buf = torch.zeros((R, C), device=t.device, dtype=t.dtype)
    buf[:rows, :cols].copy_(t)
    return buf[:, :cols] if k_rows else buf


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
    # the kernels load 16-byte (8-element) row pieces: operands of a model whose widths are not multiples of 8 are copied once into
    # zero-padded buffers (zero K-columns add nothing); a padded M (A stored transposed) or N (B stored transposed) is computed into a
    # padded C and the valid block copied out
    K8 = (K + 7) // 8 * 8
    M8, N8 = ((M + 7) // 8 * 8 if ta else M), ((N + 7) // 8 * 8 if tb else N)
    a, b = _fit8(a, ta, K8), _fit8(b, tb, K8)
    if (M8, N8) != (M, N):   # rare (widths off the 8-grid): padded product, then the epilogue terms on the valid block
        a = torch.nn.functional.pad(a, (0, M8 - a.shape[1])) if ta and a.shape[1] != M8 else a
        b = torch.nn.functional.pad(b, (0, N8 - b.shape[1])) if tb and b.shape[1] != N8 else b
        pb = torch.nn.functional.pad(bias, (0, N8 - N)) if bias is not None and N8 != N else bias
        res = gemm(a, b, ta=ta, tb=tb, out_dtype=F32, bias=pb, alpha=alpha)[:M, :N]
        if rowmask is not None:
            res = res * rowmask.reshape(-1, 1).to(F32)
        if residual is not None:
            res = res + _rows2d(residual)
        if accumulate:
            out.add_(res)
        else:
            out.copy_(res)
        return out
    K = K8
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
    ws, ws_bytes = None, _gemm_ws_bytes(M, N, K, flags)
    if ws_bytes:   # split-K partial products: the workspace is the caller's (PyTorch's caching allocator, stream-ordered reuse)
        ws = torch.empty(ws_bytes, device=a.device, dtype=torch.uint8)
    call("spn_gemm_bf16", ptr(a), ptr(b), ptr(out), ptr(bias), ptr(residual), ptr(rowmask), c_int(M), c_int(N), c_int(K),
         c_int(a.stride(0)), c_int(b.stride(0)), c_int(out.stride(0)), c_int(residual.stride(0) if residual is not None else 0),
         c_float(alpha), c_int(flags), c_int(1), c_long(0), c_long(0), c_long(0), ptr(ws), ctypes.c_size_t(ws_bytes), stream_ptr())
    return out


_WS_CACHE = {}


def _gemm_ws_bytes(M: int, N: int, K: int, flags: int) -> int:
    """spn_gemm_workspace_bytes, memoised per shape (a pure function of the shape and the split knobs)."""
    key = (M, N, K, flags & 4, _lib_mod.TUNING_EPOCH)
    n = _WS_CACHE.get(key)
    if n is None:
        n = int(load().spn_gemm_workspace_bytes(c_int(M), c_int(N), c_int(K), c_int(flags), c_int(1)))
        _WS_CACHE[key] = n
    return n


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
        if gb.dtype not in (F32, BF16) or gb.shape != (T, 2 * D):
            raise SpnError("layernorm: gb must be fp32 or bf16 [T, 2D]")
        if gb.dtype == BF16:
            call("spn_layernorm_fwd_gb16", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(gb), c_long(gb.stride(0)), ptr(y), c_int(_dt(y)),
                 c_long(y.stride(0)), ptr(mean), ptr(rstd), c_int(T), c_int(D), c_float(eps), stream_ptr())
            return y, mean, rstd
    call("spn_layernorm_fwd", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(gamma), ptr(beta), ptr(gb),
         c_long(gb.stride(0) if gb is not None else 0), ptr(y), c_int(_dt(y)), c_long(y.stride(0)), ptr(mean), ptr(rstd),
         c_int(T), c_int(D), c_float(eps), stream_ptr())
    return y, mean, rstd


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: Optional[torch.Tensor], gb: Optional[torch.Tensor],
                  mean: torch.Tensor, rstd: torch.Tensor, *, dres: Optional[torch.Tensor] = None, dx_dtype=F32,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None, want_dgb: bool = False,
                  want_dx16: bool = False, dgb_colsum: Optional[torch.Tensor] = None):
    """Returns (dx, dgb).  dgamma/dbeta (fp32 [D]) are accumulated in place when given.  `want_dx16`: the kernel also writes
    a bf16 copy of an fp32 dx, attached as `dx._spn_bf16` (the operand of the next backward GEMM; saves its cast pass).
    `dgb_colsum` (bf16 gb rows + want_dgb only): fp32 [2D], the column sums of the dgb rows are accumulated into it by the same pass."""
    x2, dy2 = _rows2d(x), _rows2d(dy)
    if dy2.dtype != BF16:
        raise SpnError("layernorm_bwd: dy must be bf16")
    T, D = x2.shape
    dx = torch.empty((T, D), device=x.device, dtype=dx_dtype)
    dgb = torch.empty((T, 2 * D), device=x.device, dtype=BF16) if want_dgb else None
    dx16 = torch.empty((T, D), device=x.device, dtype=BF16) if (want_dx16 and dx_dtype == F32) else None
    if gb is not None:
        gb = _rows2d(gb)
    if dres is not None:
        dres = _rows2d(dres)
    if gb is not None and gb.dtype == BF16 and dgb_colsum is not None:
        if dgb is None or dgb_colsum.dtype != F32 or dgb_colsum.numel() != 2 * D or not dgb_colsum.is_contiguous():
            raise SpnError("layernorm_bwd: dgb_colsum needs want_dgb and a contiguous fp32 [2D] target")
        call("spn_layernorm_bwd_gb16_colsum", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(dy2), c_long(dy2.stride(0)), ptr(gb),
             c_long(gb.stride(0)), ptr(mean), ptr(rstd), ptr(dres), c_long(dres.stride(0) if dres is not None else 0), ptr(dx),
             c_int(_dt(dx)), c_long(dx.stride(0)), ptr(dx16), c_long(D), ptr(dgb), c_long(2 * D), ptr(dgb_colsum), c_int(T), c_int(D),
             stream_ptr())
        if dx16 is not None:
            dx._spn_bf16, dx._spn_bf16_ver = dx16, dx._version
        return dx, dgb
    if gb is not None and gb.dtype == BF16:
        call("spn_layernorm_bwd_gb16", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(dy2), c_long(dy2.stride(0)), ptr(gb),
             c_long(gb.stride(0)), ptr(mean), ptr(rstd), ptr(dres), c_long(dres.stride(0) if dres is not None else 0), ptr(dx),
             c_int(_dt(dx)), c_long(dx.stride(0)), ptr(dx16), c_long(D), ptr(dgb), c_long(2 * D), c_int(T), c_int(D), stream_ptr())
        if dx16 is not None:
            dx._spn_bf16, dx._spn_bf16_ver = dx16, dx._version
        return dx, dgb
    call("spn_layernorm_bwd", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(dy2), c_long(dy2.stride(0)), ptr(gamma),
         ptr(gb), c_long(gb.stride(0) if gb is not None else 0), ptr(mean), ptr(rstd), ptr(dres),
         c_long(dres.stride(0) if dres is not None else 0), ptr(dx), c_int(_dt(dx)), c_long(dx.stride(0)), ptr(dx16), c_long(D),
         ptr(dgamma), ptr(dbeta), ptr(dgb), c_long(2 * D), c_int(T), c_int(D), stream_ptr())
    if dx16 is not None:
        dx._spn_bf16, dx._spn_bf16_ver = dx16, dx._version   # functional._bf16_grad trusts the copy at this version only
    return dx, dgb


# ---------------------------------------------------------------------------------------------------------
# attention core
# ---------------------------------------------------------------------------------------------------------

def adaln_ok(D: int, C: int) -> bool:
    return bool(load().spn_adaln_ok(c_int(D), c_int(C)))


def adaln_fwd(x: torch.Tensor, cond: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, want_gamma: bool = True):
    """y (bf16) = (W_gamma cond + b_gamma) * LN(x) + (W_beta cond + b_beta) with the projection inside the kernel; returns y, mean, rstd and
    the bf16 gamma rows [T, D] (what the backward needs of the projection; None without `want_gamma`)."""
    require_gpu(x, cond, w, bias)
    x2, c2 = _rows2d(x), _rows2d(cond)
    T, D = x2.shape
    C = c2.shape[1]
    assert x2.dtype == F32 and c2.dtype == BF16 and w.dtype == BF16 and bias.dtype == F32 and w.shape == (2 * D, C) and w.is_contiguous()
    assert x2.stride(1) == 1 and c2.stride(1) == 1 and c2.shape[0] == T
    y = torch.empty((T, D), device=x.device, dtype=BF16)
    gam = torch.empty((T, D), device=x.device, dtype=BF16) if want_gamma else None
    mean = torch.empty(T, device=x.device, dtype=F32)
    rstd = torch.empty(T, device=x.device, dtype=F32)
    call("spn_adaln_fwd", ptr(x2), c_long(x2.stride(0)), ptr(c2), c_long(c2.stride(0)), ptr(w), ptr(bias), ptr(y), c_long(D), ptr(gam), c_long(D),
         ptr(mean), ptr(rstd), c_int(T), c_int(D), c_int(C), c_float(eps), stream_ptr())
    return y, mean, rstd, gam


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


def attn_fwd(q, k, v, *, kmask=None, qmask=None, slopes=None, causal=False, scale=None, p_drop: float = 0.0, seed: int = 0, band=None):
    """q [b,nq,h,64], k/v [b,nk,kvh,64] (kvh = 1 or h), kmask [b,nk] bool, slopes [h] fp32 -> o [b,nq,h,64], lse [b,h,nq].

    With p_drop > 0 a third value is returned: the dropout keep bits (int16 words, 1 bit per score) that `attn_bwd` needs.
    `band`: fp32 buffer from `attn_band_buffer` that receives the ALiBi band bounds, for `attn_bwd` to reuse; with slopes and
    no buffer one is allocated here (the library owns no memory), without slopes there is no band.
    `qmask` [b,nq] bool: the query-side padding mask -- rows with False come back as zeros (the module multiplies its output by
    this mask anyway, attention.py:216-218) with a dead lse, and blocks of such rows are not computed (include/spn.h)."""
    if band is None and slopes is not None:
        band = attn_band_buffer(q, k)
    require_gpu(q, k, v)
    b, nq, h, dh = q.shape
    nk, kvh = k.shape[1], k.shape[2]
    o = torch.empty((b, nq, h, dh), device=q.device, dtype=BF16)
    lse = torch.empty((b, h, nq), device=q.device, dtype=F32)
    ks, vs = _bnhd_strides(k), _bnhd_strides(v)
    if kvh == 1:
        ks, vs = (ks[0], ks[1], 0), (vs[0], vs[1], 0)
    strides = (c_long * 12)(*_bnhd_strides(q), *ks, *vs, *_bnhd_strides(o))
    kmask, qmask = _mask_u8(kmask), _mask_u8(qmask)
    if qmask is not None and tuple(qmask.shape) != (b, nq):
        raise SpnError(f"attn_fwd: qmask must be [b, nq] = {(b, nq)}, got {tuple(qmask.shape)}")
    bits = None
    if p_drop > 0:
        bits = torch.empty(load().spn_attn_dropbits_elems(c_int(b), c_int(h), c_int(nq), c_int(nk)), device=q.device, dtype=torch.int16)
    call("spn_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(kmask), ptr(qmask), ptr(slopes), c_int(b), c_int(h), c_int(kvh),
         c_int(nq), c_int(nk), c_int(1 if causal else 0), c_float(scale if scale is not None else dh ** -0.5), strides,
         c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), ptr(bits), ptr(band), stream_ptr())
    return (o, lse) if bits is None else (o, lse, bits)


def attn_band_buffer(q, k) -> torch.Tensor:
    """Buffer for the band bounds of a [b,nq,h,64] x [b,nk,kvh,64] attention (filled by attn_fwd, reused by attn_bwd)."""
    b, nq, h, _ = q.shape
    n = load().spn_attn_band_elems(c_int(b), c_int(h), c_int(k.shape[2]), c_int(nq))
    return torch.empty(n, device=q.device, dtype=F32)


def attn_set_band(log2_threshold: float) -> None:
    """ALiBi band skipping threshold (log2 of the smallest probability ratio still visited); 0 disables (tests, ablations)."""
    load().spn_attn_set_band(c_float(log2_threshold))


def attn_bwd(q, k, v, o, d_o, lse, *, dq, dk, dv, kmask=None, qmask=None, slopes=None, causal=False, scale=None, want_dslope=False,
             p_drop: float = 0.0, dropbits=None, band=None):
    """Writes dq/dk/dv ([b,n,h|kvh,64] bf16 views, e.g. slices of a fused dqkv buffer); returns dslope [h] fp32 or None.
    `dropbits`: the keep bits returned by `attn_fwd` when p_drop > 0; `qmask`: the forward's (its padding rows get dq = 0)."""
    if p_drop > 0 and dropbits is None:
        raise SpnError("attn_bwd: p_drop > 0 needs the keep bits of the forward")
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
    dslope = zeros_small(h, q.device) if want_dslope else None
    kmask, qmask = _mask_u8(kmask), _mask_u8(qmask)
    call("spn_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
         ptr(dslope), ptr(kmask), ptr(qmask), ptr(slopes), c_int(b), c_int(h), c_int(kvh), c_int(nq), c_int(nk),
         c_int(1 if causal else 0), c_float(scale if scale is not None else dh ** -0.5), strides, c_float(p_drop),
         ptr(dropbits), ptr(band), stream_ptr())
    return dslope


# ---------------------------------------------------------------------------------------------------------
# casts / element-wise
# ---------------------------------------------------------------------------------------------------------

def _bt_view(t: torch.Tensor):
    """Return (tensor, B, t_len, D, batch_stride, row_stride) for a [.., D] view addressable as (batch, row)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    if t.ndim == 2:
        return t, 1, t.shape[0], t.shape[1], 0, t.stride(0)
    if t.ndim == 3:
        return t, t.shape[0], t.shape[1], t.shape[2], t.stride(0), t.stride(1)
    t = t.reshape(-1, t.shape[-1])
    return t, 1, t.shape[0], t.shape[1], 0, t.stride(0)


def cast(x: torch.Tensor, dtype, *, rowmask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[..., :] = dtype(x[..., :]) * rowmask[...]; x may be a strided [b, t, D] view, out a slice of a wider buffer."""
    require_gpu(x)
    x, B, t_len, D, xbs, xts = _bt_view(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    o, Bo, to, Do, obs, ots = _bt_view(out)
    if (Bo * to, Do) != (B * t_len, D):
        raise SpnError("cast: shape mismatch")
    if Bo != B:  # address the output with the input's (batch,row) split
        if o.ndim == 2:
            obs, ots = t_len * o.stride(0), o.stride(0)
        else:
            raise SpnError("cast: incompatible batch split")
    if rowmask is not None:
        rowmask = _mask_u8(rowmask.reshape(-1))
    call("spn_cast", ptr(x), c_int(_dt(x)), c_long(xbs), c_long(xts), ptr(o), c_int(_dt(o)), c_long(obs), c_long(ots),
         ptr(rowmask), c_long(B), c_long(t_len), c_int(D), stream_ptr())
    return out


def act_fwd(u: torch.Tensor, *, act: int, glu: bool, p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    u2 = _rows2d(u)
    T, W = u2.shape
    I = W // 2 if glu else W
    out = torch.empty((T, I), device=u.device, dtype=BF16)
    call("spn_act_fwd", ptr(u2), c_long(u2.stride(0)), ptr(out), c_long(I), c_long(T), c_int(I), c_int(act), c_int(int(glu)),
         c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), stream_ptr())
    return out


def gemm_glu_ok(M: int, I: int, K: int) -> bool:
    return bool(load().spn_gemm_glu_ok(c_int(M), c_int(I), c_int(K)))


def gemm_glu(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, act: int, p_drop: float = 0.0, seed: int = 0):
    """(u, g): u = x @ w.T + bias as bf16 [M, 2I]; g = dropout(u[:, :I] * act(u[:, I:])) written by the same kernel."""
    M, K = x.shape
    I = w.shape[0] // 2
    assert x.dtype == BF16 and w.dtype == BF16 and x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    u = torch.empty((M, 2 * I), device=x.device, dtype=BF16)
    g = torch.empty((M, I), device=x.device, dtype=BF16)
    b = bias if bias is None or bias.dtype == F32 else bias.float()
    call("spn_gemm_glu", ptr(x), ptr(w), ptr(u), ptr(g), ptr(b), c_int(M), c_int(I), c_int(K), c_int(x.stride(0)), c_int(w.stride(0)),
         c_int(2 * I), c_int(I), c_int(act), c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), stream_ptr())
    return u, g


def gemm_glu_bwd_ok(M: int, I: int, K: int) -> bool:
    return bool(load().spn_gemm_glu_bwd_ok(c_int(M), c_int(I), c_int(K)))


def gemm_glu_bwd(dy: torch.Tensor, w2: torch.Tensor, u: torch.Tensor, *, act: int, p_drop: float = 0.0, seed: int = 0,
                 colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """du [M, 2I] = act_bwd(u, dy @ w2) in ONE kernel (w2: the output projection's weight [K, I]; dg is never stored).
    `colsum` (fp32 [2I]): the column sums of du are accumulated into it (bias gradient of the input projection)."""
    M, K = dy.shape
    I = w2.shape[1]
    assert dy.dtype == BF16 and w2.dtype == BF16 and u.dtype == BF16 and dy.stride(1) == 1 and w2.stride(1) == 1 and u.stride(1) == 1
    assert w2.shape[0] == K and u.shape == (M, 2 * I)
    du = torch.empty((M, 2 * I), device=dy.device, dtype=BF16)
    part = torch.empty(((M + 127) // 128, 2 * I), device=dy.device, dtype=F32) if colsum is not None else None
    call("spn_gemm_glu_bwd", ptr(dy), ptr(w2), ptr(u), ptr(du), ptr(part), c_int(M), c_int(I), c_int(K), c_int(dy.stride(0)),
         c_int(w2.stride(0)), c_int(u.stride(0)), c_int(2 * I), c_int(act), c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), stream_ptr())
    if colsum is not None:
        call("spn_colsum", ptr(part), c_int(0), c_long(2 * I), ptr(colsum), c_long(part.shape[0]), c_int(2 * I), stream_ptr())
    return du


def act_bwd(u: torch.Tensor, dout: torch.Tensor, *, act: int, glu: bool, p_drop: float = 0.0, seed: int = 0,
            colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`colsum` (fp32 [W]): the column sums of the result are accumulated into it (bias gradient of the Linear that produced u)."""
    u2, d2 = _rows2d(u), _rows2d(dout)
    T, W = u2.shape
    I = W // 2 if glu else W
    du = torch.empty((T, W), device=u.device, dtype=BF16)
    ws = torch.empty(4096 * W, device=u.device, dtype=F32) if colsum is not None else None   # per-block partial rows
    call("spn_act_bwd", ptr(u2), c_long(u2.stride(0)), ptr(d2), c_long(d2.stride(0)), ptr(du), c_long(W), c_long(T), c_int(I),
         c_int(act), c_int(int(glu)), c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), ptr(colsum), ptr(ws), stream_ptr())
    return du


def act_bwd_can_fuse_colsum(width: int, glu: bool) -> bool:
    chunks = (width // 2 if glu else width) // 8
    return chunks > 0 and (chunks % 256 == 0 or 256 % chunks == 0)


def dropout(x: torch.Tensor, p_drop: float, seed: int) -> torch.Tensor:
    """Stand-alone dropout of [.., D] (fp32 or bf16, unit inner stride; D a multiple of 8 runs row-wise -- a row stride off the 8-element grid is
    packed first --, other widths through a padded flat copy): kept entries scaled by 1 / (1 - p), mask a
    function of (seed, row, column) -- calling it again on the gradient with the same seed is the backward."""
    require_gpu(x)
    x2 = _rows2d(x)
    if x2.dtype not in (F32, BF16):
        raise SpnError("dropout: fp32 / bf16 input")
    if x2.shape[1] % 8:
        # a width off the kernel's 8-element grid (the reference's nn.Dropout takes any): the same kernel over the tensor laid out as
        # zero-padded rows of 256 -- the mask is then a function of (seed, flat index), which the backward reproduces on a gradient of the
        # same shape.  One pad copy and one slice; no shipped recipe comes here.  The choice depends on the WIDTH only, so the forward (on
        # x) and the backward (on dy, whatever its strides) always agree on the mask's definition.
        n = x.numel()
        flat = torch.zeros(((n + 255) // 256) * 256, device=x.device, dtype=x.dtype)
        flat[:n] = x.reshape(-1)
        return dropout(flat.view(-1, 256), p_drop, seed).view(-1)[:n].view(x.shape)
    if x2.stride(0) % 8 or x2.data_ptr() % 16:      # a narrowed view whose rows start off the grid: same (seed, row, column) mask over a packed copy
        x2 = x2.contiguous()
    y = torch.empty(x2.shape, device=x.device, dtype=x2.dtype)
    call("spn_dropout", ptr(x2), c_long(x2.stride(0)), ptr(y), c_long(y.stride(0)), c_int(0 if x2.dtype == F32 else 1), c_long(x2.shape[0]),
         c_int(x2.shape[1]), c_float(p_drop), ctypes.c_uint(seed & 0xFFFFFFFF), stream_ptr())
    return y.view(x.shape)


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[N] (fp32) += column sums of x [T, N]."""
    x2 = _rows2d(x)
    T, N = x2.shape
    if out is None:
        out = zeros_small(N, x.device)
    call("spn_colsum", ptr(x2), c_int(_dt(x2)), c_long(x2.stride(0)), ptr(out), c_long(T), c_int(N), stream_ptr())
    return out


def mish_fwd(x: torch.Tensor) -> torch.Tensor:
    """y = x * tanh(softplus(x)), fp32 (nn.Mish of the embedding value MLP and of the embedding head)."""
    require_gpu(x)
    if x.dtype != F32:
        raise SpnError("mish: fp32 only")
    x = x.contiguous()
    y = torch.empty_like(x)
    call("spn_mish_fwd", ptr(x), ptr(y), c_long(x.numel()), stream_ptr())
    return y


def mish_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    x, dy = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    call("spn_mish_bwd", ptr(x), ptr(dy), ptr(dx), c_long(x.numel()), stream_ptr())
    return dx


def gemm_f32(a: torch.Tensor, b: torch.Tensor, *, ta=False, tb=False, bias=None, rowmask=None, out=None, alpha=1.0,
             accumulate=False) -> torch.Tensor:
    """Exact-fp32 C = alpha * op(A) @ op(B) + bias.  a: [M,K] (or [K,M] if ta), b: [N,K] nn.Linear-style (or [K,N] if tb)."""
    require_gpu(a, b)
    if a.dtype != F32 or b.dtype != F32:
        raise SpnError("gemm_f32 operands must be fp32")
    if a.ndim != 2:
        a = a.reshape(-1, a.shape[-1])
    if b.ndim != 2:
        b = b.reshape(-1, b.shape[-1])
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    if K != Kb:
        raise SpnError("gemm_f32: inner dimensions differ")
    sam, sak = (a.stride(1), a.stride(0)) if ta else (a.stride(0), a.stride(1))
    sbk, sbn = (b.stride(0), b.stride(1)) if tb else (b.stride(1), b.stride(0))
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=F32)
    if rowmask is not None:
        rowmask = _mask_u8(rowmask.reshape(-1))
    call("spn_gemm_f32", ptr(a), c_long(sam), c_long(sak), ptr(b), c_long(sbk), c_long(sbn), ptr(out), c_long(out.stride(0)),
         ptr(bias), ptr(rowmask), c_int(M), c_int(N), c_int(K), c_float(alpha), c_int(int(accumulate)), stream_ptr())
    return out


def zero_masked_rows(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """In place: bf16 rows of x [R, D] with m[r] == 0 become zero (only those rows are written)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != BF16 or x2.stride(1) != 1:
        raise SpnError("zero_masked_rows: bf16 rows with unit inner stride")
    call("spn_zero_masked_rows", ptr(x2), c_long(x2.stride(0)), ptr(_mask_u8(m.reshape(-1))), c_long(x2.shape[0]), c_int(x2.shape[1]),
         stream_ptr())
    return x


def rows_all_nonzero(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    m = torch.empty(x2.shape[0], device=x.device, dtype=torch.uint8)
    call("spn_rows_all_nonzero", ptr(x2), c_long(x2.stride(0)), ptr(m), c_long(x2.shape[0]), c_int(x2.shape[1]), stream_ptr())
    return m.view(torch.bool).view(x.shape[:-1])


def mask_rows(x: torch.Tensor, m: torch.Tensor, invert: bool = False) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    y = torch.empty_like(x2)
    call("spn_mask_rows", ptr(x2), c_long(x2.stride(0)), ptr(_mask_u8(m.reshape(-1))), ptr(y), c_long(y.stride(0)),
         c_long(x2.shape[0]), c_int(x2.shape[1]), c_int(int(invert)), stream_ptr())
    return y.view(x.shape)


# ---------------------------------------------------------------------------------------------------------
# embedding tables + tuple gather
# ---------------------------------------------------------------------------------------------------------

def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


def table_build_fwd(tv, w0, b0, w1, b1, iw, *, dense: bool, discrete: bool, ids_mask: int):
    """Per-key lists of fp32 tensors -> (tables [list of [V,E]], h1 [list])."""
    n = len(tv)
    V = [t.numel() for t in tv]
    E = [w.numel() for w in w0]
    dev = tv[0].device
    buf = torch.empty(sum(v * e for v, e in zip(V, E)), device=dev, dtype=F32)
    hbuf = torch.empty_like(buf) if dense else None
    outs, h1s, off = [], [], 0
    for v, e in zip(V, E):
        outs.append(buf[off:off + v * e].view(v, e))
        h1s.append(hbuf[off:off + v * e].view(v, e) if dense else None)
        off += v * e
    none = [None] * n
    call("spn_table_build_fwd", c_int(n), _ptr_array(tv), _ptr_array(w0), _ptr_array(b0 if dense else none),
         _ptr_array(w1 if dense else none), _ptr_array(b1 if dense else none), _ptr_array(iw if iw is not None else none),
         _ptr_array(outs), _ptr_array(h1s), _int_array(V), _int_array(E), c_int(int(dense)), c_int(int(discrete)),
         ctypes.c_uint(ids_mask), stream_ptr())
    return outs, h1s


def table_build_bwd(tv, w0, b0, w1, dout, h1, *, has_iw: bool, dense: bool, discrete: bool, ids_mask: int):
    """Returns per-key grads (dw0, db0, dw1, db1, diw)."""
    n = len(tv)
    V = [t.numel() for t in tv]
    E = [w.numel() for w in w0]
    dev = tv[0].device
    dout = [d.contiguous() for d in dout]
    # one zero fill for all the accumulated vectors (3 per key), handed out as views
    zbuf = torch.zeros((3 if dense else 1) * sum(E), device=dev, dtype=F32)
    offs = [0]
    for e in E:
        offs.append(offs[-1] + e)
    dw0 = [zbuf[offs[i]:offs[i + 1]] for i in range(n)]
    db0 = [zbuf[offs[-1] + offs[i]:offs[-1] + offs[i + 1]] for i in range(n)] if dense else [None] * n
    db1 = [zbuf[2 * offs[-1] + offs[i]:2 * offs[-1] + offs[i + 1]] for i in range(n)] if dense else [None] * n
    diw = [torch.empty(v, e, device=dev, dtype=F32) for v, e in zip(V, E)] if has_iw else [None] * n
    dval = [torch.empty(v, e, device=dev, dtype=F32) for v, e in zip(V, E)] if dense else [None] * n
    none = [None] * n
    call("spn_table_build_bwd", c_int(n), _ptr_array(tv), _ptr_array(w0), _ptr_array(b0 if dense else none),
         _ptr_array(w1 if dense else none), _ptr_array(dout), _ptr_array(dw0), _ptr_array(db0), _ptr_array(db1),
         _ptr_array(diw), _ptr_array(dval), _int_array(V), _int_array(E), c_int(int(dense)), c_int(int(discrete)),
         ctypes.c_uint(ids_mask), stream_ptr())
    dw1 = [None] * n
    if dense:  # dW1[e, j] = sum_v dval[v, e] * h1[v, j]
        dw1 = [gemm_f32(dval[i], h1[i], ta=True, tb=True) for i in range(n)]
    return dw0, db0, dw1, db1, diw


def _tok_view(tokens: torch.Tensor):
    if tokens.dtype != torch.int64 or tokens.stride(-1) != 1 or tokens.ndim != 3:
        tokens = tokens.long().contiguous()
        if tokens.ndim == 2:
            tokens = tokens[None]
    return tokens, tokens.shape[0], tokens.shape[1], tokens.stride(0), tokens.stride(1)


def embed_fwd(tables, tokens, gamma, beta, eps: float = 1e-5):
    """tokens int64 [b, t, >=K] (any batch/row strides) -> (y bf16 [b*t, sum E], mean, rstd)."""
    tokens, B, t_len, tbs, tts = _tok_view(tokens)
    T = B * t_len
    V = [t.shape[0] for t in tables]
    E = [t.shape[1] for t in tables]
    D = sum(E)
    dev = tables[0].device
    y = torch.empty((T, D), device=dev, dtype=BF16)
    mean = torch.empty(T, device=dev, dtype=F32) if gamma is not None else None
    rstd = torch.empty(T, device=dev, dtype=F32) if gamma is not None else None
    call("spn_embed_fwd", c_int(len(tables)), _ptr_array(tables), _int_array(V), _int_array(E), ptr(tokens), c_long(tbs),
         c_long(tts), c_int(t_len), ptr(gamma), ptr(beta), ptr(y), c_long(D), ptr(mean), ptr(rstd), c_int(T), c_float(eps),
         stream_ptr())
    return y, mean, rstd


def embed_bwd(tables, tokens, dy, gamma, mean, rstd, *, dgamma=None, dbeta=None, padding_idx: int = 0, out=None):
    """Returns list of dtables (fp32, fresh -- or `out`: contiguous fp32 [V, E] tensors the kernel ADDS to); dgamma/dbeta accumulated in place."""
    tokens, B, t_len, tbs, tts = _tok_view(tokens)
    T = B * t_len
    V = [t.shape[0] for t in tables]
    E = [t.shape[1] for t in tables]
    dev = tables[0].device
    dy2 = _rows2d(dy)
    if out is not None:
        dts = list(out)
        for d, v, e in zip(dts, V, E):
            if d.dtype != F32 or tuple(d.shape) != (v, e) or not d.is_contiguous():
                raise SpnError("embed_bwd: out tables must be contiguous fp32 [V, E]")
    else:
        buf = torch.zeros(sum(v * e for v, e in zip(V, E)), device=dev, dtype=F32)
        dts, off = [], 0
        for v, e in zip(V, E):
            dts.append(buf[off:off + v * e].view(v, e))
            off += v * e
    ws = torch.empty(2 * T, device=dev, dtype=F32) if gamma is not None else None
    call("spn_embed_bwd", c_int(len(tables)), _ptr_array(tables), _ptr_array(dts), _int_array(V), _int_array(E), ptr(tokens),
         c_long(tbs), c_long(tts), c_int(t_len), ptr(dy2), c_long(dy2.stride(0)), ptr(gamma), ptr(mean), ptr(rstd), ptr(dgamma),
         ptr(dbeta), ptr(ws), c_int(T), c_int(padding_idx), stream_ptr())
    return dts


# ---------------------------------------------------------------------------------------------------------
# losses, segments, MMD
# ---------------------------------------------------------------------------------------------------------

def _label_view(labels: torch.Tensor):
    """labels: int64 [b, t] view (possibly a strided slice labels[:, 1:, i])."""
    if labels.dtype != torch.int64:
        labels = labels.long()
    if labels.ndim == 1:
        labels = labels[None]
    return labels, labels.shape[0], labels.shape[1], labels.stride(0), labels.stride(1)


def ce_fwd(logits: torch.Tensor, V: int, labels: torch.Tensor, *, ignore_index: int = -100, want_argmax: bool = False,
           eval_spec=None):
    """logits [T, >=V] (row stride free) -> (lse [T], sums [2] = (loss sum, valid count), argmax int32 [T] | None).
    eval_spec = (token_values fp32 [V] | None, weighted): also returns metrics [2] = (#correct, distance sum) from the same pass."""
    lg = _rows2d(logits)
    labels, B, t_len, lbs, lts = _label_view(labels)
    T = B * t_len
    if lg.shape[0] != T:
        raise SpnError("ce_fwd: logits rows != labels")
    lse = torch.empty(T, device=lg.device, dtype=F32)
    sums = zeros_small(2, lg.device)
    am = torch.empty(T, device=lg.device, dtype=torch.int32) if want_argmax else None
    if eval_spec is not None:
        tv, weighted = eval_spec
        if tv is not None and (tv.dtype != F32 or tv.numel() < V or not tv.is_contiguous() or tv.device != lg.device):
            raise SpnError("ce_fwd: token values must be a contiguous fp32 [V] tensor on the logits' device")
        metrics = zeros_small(2, lg.device)
        call("spn_ce_fwd_eval", ptr(lg), c_int(_dt(lg)), c_long(lg.stride(0)), ptr(labels), c_long(lbs), c_long(lts), c_int(t_len),
             c_int(ignore_index), ptr(lse), ptr(sums), ptr(am), ptr(tv), c_int(int(bool(weighted))), ptr(metrics), c_long(T), c_int(V),
             stream_ptr())
        return lse, sums, am, metrics
    call("spn_ce_fwd", ptr(lg), c_int(_dt(lg)), c_long(lg.stride(0)), ptr(labels), c_long(lbs), c_long(lts), c_int(t_len),
         c_int(ignore_index), ptr(lse), ptr(sums), ptr(am), c_long(T), c_int(V), stream_ptr())
    return lse, sums, am


def ce_bwd(logits: torch.Tensor, V: int, labels: torch.Tensor, lse: torch.Tensor, coef: torch.Tensor, *, ignore_index=-100,
           Vpad: Optional[int] = None) -> torch.Tensor:
    """dlogits bf16 [T, Vpad] = coef * (softmax - onehot) on non-ignored rows (pad columns zero)."""
    lg = _rows2d(logits)
    labels, B, t_len, lbs, lts = _label_view(labels)
    T = B * t_len
    Vpad = Vpad or ((V + 7) // 8) * 8
    dl = torch.empty((T, Vpad), device=lg.device, dtype=BF16)
    call("spn_ce_bwd", ptr(lg), c_int(_dt(lg)), c_long(lg.stride(0)), ptr(labels), c_long(lbs), c_long(lts), c_int(t_len),
         c_int(ignore_index), ptr(lse), ptr(coef), ptr(dl), c_long(Vpad), c_long(T), c_int(V), c_int(Vpad), stream_ptr())
    return dl


def segment_count(seg: torch.Tensor, S: int) -> torch.Tensor:
    seg = seg.contiguous()
    b, t = seg.shape
    counts = torch.zeros((b, S), device=seg.device, dtype=F32)
    call("spn_segment_count", ptr(seg), ptr(counts), c_int(b), c_int(t), c_int(S), stream_ptr())
    return counts


def segment_sum(x: torch.Tensor, seg: torch.Tensor, S: int, *, counts: Optional[torch.Tensor] = None,
                rowmask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [b,t,d] (fp32/bf16, unit inner stride) -> out fp32 [b,S,d]: per-segment sums, or means when counts is given.  `out` may be a
    column slice `agg[..., c0:c0+d]` of a contiguous [b,S,W] buffer; sums are ADDED to it, means need it ZERO on entry (a run that holds
    every row of its id is stored, not added)."""
    seg = seg.contiguous()
    if x.stride(-1) != 1:
        x = x.contiguous()
    b, t, d = x.shape
    if out is None:
        out = torch.zeros((b, S, d), device=x.device, dtype=F32)
    elif out.stride(-1) != 1 or out.stride(0) != S * out.stride(1) or tuple(out.shape) != (b, S, d) or out.dtype != F32:
        raise SpnError("segment_sum: output must be an fp32 column slice of a contiguous [b,S,W] buffer")
    if rowmask is not None:
        rowmask = _mask_u8(rowmask.reshape(-1))
    call("spn_segment_sum", ptr(x), c_int(_dt(x)), c_long(x.stride(0)), c_long(x.stride(1)), ptr(seg), ptr(counts), ptr(rowmask),
         ptr(out), c_long(out.stride(1)), c_int(b), c_int(t), c_int(S), c_int(d), stream_ptr())
    return out


def _level_arrays(segs, counts, bufs, S):
    nl = len(segs)
    keep = [s.contiguous() for s in segs]
    if any(s.dtype != torch.int64 for s in keep):
        raise SpnError("segment ids must be int64")
    for bf, s_ in zip(bufs, S):
        if bf.dtype != F32 or bf.stride(-1) != 1 or bf.stride(0) != s_ * bf.stride(1):
            raise SpnError("per-level buffers must be fp32 column slices of contiguous [b,S,W] buffers")
    a_seg = (ctypes.c_void_p * nl)(*[s.data_ptr() for s in keep])
    a_cnt = (ctypes.c_void_p * nl)(*[c.data_ptr() for c in counts])
    a_buf = (ctypes.c_void_p * nl)(*[bf.data_ptr() for bf in bufs])
    a_ld = (ctypes.c_long * nl)(*[bf.stride(1) for bf in bufs])
    a_S = (ctypes.c_int * nl)(*[int(v) for v in S])
    return keep, a_seg, a_cnt, a_buf, a_ld, a_S


def segment_sum_multi(x: torch.Tensor, rowmask: Optional[torch.Tensor], segs, counts, outs, S) -> None:
    """All levels in one pass: outs[l][b, S_l, :d] (fp32, zeroed by the caller, may be column slices) += segment means of
    x[b, t, :d] * rowmask under segs[l] (counts[l] from segment_count)."""
    require_gpu(x)
    if x.stride(-1) != 1:
        x = x.contiguous()
    b, t, d = x.shape
    keep, a_seg, a_cnt, a_out, a_ld, a_S = _level_arrays(segs, counts, outs, S)
    rm = None if rowmask is None else _mask_u8(rowmask.reshape(-1))
    call("spn_segment_sum_multi", ptr(x), c_int(_dt(x)), c_long(x.stride(0)), c_long(x.stride(1)), ptr(rm), c_int(len(segs)), a_seg, a_cnt,
         a_out, a_ld, a_S, c_int(b), c_int(t), c_int(d), stream_ptr())


def segment_gather_multi(srcs, segs, counts, S, rowmask: Optional[torch.Tensor], d: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[b, t, :d] = rowmask * sum_l srcs[l][b, segs[l][b, t], :d] / max(counts[l], 1): the backward of segment_sum_multi, written once."""
    b, t = segs[0].shape
    if out is None:
        out = torch.empty((b, t, d), device=srcs[0].device, dtype=F32)
    if out.stride(-1) != 1 or out.stride(0) != t * out.stride(1):
        raise SpnError("segment_gather_multi: output must be a column slice of a contiguous [b,t,W] buffer")
    keep, a_seg, a_cnt, a_src, a_ld, a_S = _level_arrays(segs, counts, srcs, S)
    rm = None if rowmask is None else _mask_u8(rowmask.reshape(-1))
    call("spn_segment_gather_multi", c_int(len(segs)), a_src, a_ld, a_seg, a_cnt, a_S, ptr(rm), ptr(out), c_long(out.stride(1)), c_int(b),
         c_int(t), c_int(d), stream_ptr())
    return out


def segment_gather(src: torch.Tensor, seg: torch.Tensor, *, counts: Optional[torch.Tensor] = None,
                   rowmask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                   accumulate: bool = False) -> torch.Tensor:
    """src fp32 [b,S,d] -> y fp32 [b,t,d] (+)= src[b, seg] (/counts) (*rowmask).  `out` may be a column slice
    `wide[..., c0:c0+d]` of a contiguous [b,t,W] buffer, `src` a column slice of a contiguous [b,S,W] one."""
    seg = seg.contiguous()
    b, S, d = src.shape
    if src.stride(-1) != 1 or src.stride(0) != S * src.stride(1):
        src = src.contiguous()
    t = seg.shape[1]
    if out is None:
        out = torch.empty((b, t, d), device=src.device, dtype=F32)
    if out.stride(-1) != 1 or out.stride(0) != t * out.stride(1):
        raise SpnError("segment_gather: output must be a column slice of a contiguous [b,t,W] buffer")
    if rowmask is not None:
        rowmask = _mask_u8(rowmask.reshape(-1))
    call("spn_segment_gather", ptr(src), c_long(src.stride(1)), ptr(seg), ptr(counts), ptr(rowmask), ptr(out), c_long(out.stride(1)), c_int(b), c_int(t),
         c_int(S), c_int(d), c_int(int(accumulate)), stream_ptr())
    return out


class _SmallZeros:
    """Small fp32 accumulators that have to start at zero (atomic targets, partial sums): slices of a 64 KiB block that is cleared by ONE
    fill when it is allocated, instead of one fill launch per accumulator (~80 per train step).  A slice is handed out exactly once --
    an exhausted block is dropped, never recycled -- so this is plain allocation batching: no aliasing, no lifetime rule.
    While a stream is being captured into a graph the fill has to be part of the graph (a replay must find zeros again), so a capture
    gets a tensor and a fill of its own."""
    BLOCK = 16384   # floats

    def __init__(self):
        import threading
        self.blocks = {}
        self.lock = threading.Lock()

    def take(self, n: int, device) -> torch.Tensor:
        if n > 1024 or (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            return torch.zeros(n, device=device, dtype=F32)
        step = (n + 63) // 64 * 64       # 256-byte granules keep every slice aligned for vector accesses
        # one block per (device, stream): the fill is ordered on the stream that allocates the block, and so is every later user of a slice
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        with self.lock:
            blk = self.blocks.get(key)
            if blk is None or blk[1] + step > self.BLOCK:
                blk = [torch.zeros(self.BLOCK, device=device, dtype=F32), 0]
                self.blocks[key] = blk
            # `.data`: a tensor on the same storage with a VERSION COUNTER OF ITS OWN.  Plain slices of one block would all share the
            # block's counter, and an in-place torch op on any of them (AccumulateGrad's `grad += ...` on a bias gradient that came from
            # here, clip_grad_norm_'s `mul_`) would invalidate every other slice that some autograd node has saved for its backward
            out = blk[0].data[blk[1]:blk[1] + n]
            blk[1] += step
        return out


_SMALL_ZEROS = _SmallZeros()


def zeros_small(n: int, device) -> torch.Tensor:
    return _SMALL_ZEROS.take(int(n), torch.device(device))


def mmd_scalars(sums: torch.Tensor, Z: int, g: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g None: the MMD value (0-dim) from mmd_fwd's sums; g (device scalar dL/dmmd): the [2] coefficient vector mmd_bwd takes."""
    out = torch.empty(2 if g is not None else 1, device=sums.device, dtype=F32)
    if g is not None:
        g = g.reshape(1).float().contiguous()
    call("spn_mmd_scalars", ptr(sums), c_int(int(Z)), ptr(g), ptr(out), stream_ptr())
    return out if g is not None else out[0]


def mmd_fwd(z: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """sums[4] = (sum k(z,z), sum w w k(y,y), sum w k(z,y), sum w)."""
    z, y, w = z.contiguous(), y.contiguous(), w.contiguous()
    sums = zeros_small(4, y.device)
    call("spn_mmd_fwd", ptr(z), c_int(z.shape[0]), ptr(y), ptr(w), c_int(y.shape[0]), c_int(y.shape[1]), ptr(sums), stream_ptr())
    return sums


def mmd_bwd(z: torch.Tensor, y: torch.Tensor, w: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    z, y, w = z.contiguous(), y.contiguous(), w.contiguous()
    dy = torch.empty_like(y)
    call("spn_mmd_bwd", ptr(z), c_int(z.shape[0]), ptr(y), ptr(w), c_int(y.shape[0]), c_int(y.shape[1]), ptr(coef), ptr(dy),
         stream_ptr())
    return dy


# ---------------------------------------------------------------------------------------------------------
# latent stage behind the VAE heads (csrc/latent.hip)
# ---------------------------------------------------------------------------------------------------------

LATENT_SELECT_MAX_N, LATENT_SELECT_MAX_K, LATENT_SELECT_MAX_B = 262144, 4096, 1024


def latent_select(lat: torch.Tensor, valid: torch.Tensor, deadpan: Optional[torch.Tensor], K: int, seed: int):
    """lat [b, S, D] fp32, valid [b, S] bool, deadpan [b] bool or None -> (y [K, D], w [K], slot [b * S] int32, dead [3]): a uniform
    random subset of at most K valid latents packed into y with 0 / 1 row weights, and the level's deadpan sums (see include/spn.h)."""
    require_gpu(lat)
    b, S, D = lat.shape
    lat = lat.contiguous()
    dev = lat.device
    y = torch.empty((K, D), device=dev, dtype=F32)
    w = torch.empty(K, device=dev, dtype=F32)
    slot = torch.empty(b * S, device=dev, dtype=torch.int32)
    dead = torch.empty(3, device=dev, dtype=F32)
    valid = _mask_u8(valid)
    if valid.data_ptr() % 16:          # (a view at an odd offset: the kernel reads the bytes 16 at a time)
        valid = valid.clone()
    call("spn_latent_select", ptr(lat), ptr(valid), ptr(_mask_u8(deadpan)), c_int(b * S), c_int(S), c_int(D), c_int(K),
         ctypes.c_uint(seed & 0xFFFFFFFF), ptr(y), ptr(w), ptr(slot), ptr(dead), stream_ptr())
    return y, w, slot, dead


def latent_unselect(dy: Optional[torch.Tensor], slot: torch.Tensor, lat: torch.Tensor, valid: torch.Tensor,
                    deadpan: Optional[torch.Tensor], dead: torch.Tensor, g_dead: Optional[torch.Tensor]) -> torch.Tensor:
    b, S, D = lat.shape
    lat = lat.contiguous()
    dlat = torch.empty_like(lat)
    if g_dead is not None:
        g_dead = g_dead.reshape(1).float().contiguous()
    call("spn_latent_unselect", ptr(None if dy is None else dy.contiguous()), ptr(slot), ptr(lat), ptr(_mask_u8(valid)), ptr(_mask_u8(deadpan)),
         ptr(dead), ptr(g_dead), c_int(b * S), c_int(S), c_int(D), ptr(dlat), stream_ptr())
    return dlat


def latent_scalars(sums: torch.Tensor, Z: int, dead: torch.Tensor, D: int, weight: float) -> torch.Tensor:
    out = torch.empty(3, device=sums.device, dtype=F32)
    call("spn_latent_scalars", ptr(sums), c_int(int(Z)), ptr(dead), c_int(int(D)), c_float(float(weight)), ptr(out), stream_ptr())
    return out


def latent_drop(emb: torch.Tensor, mask: torch.Tensor, deadpan: Optional[torch.Tensor], levels, inclusive: bool, seed: int):
    """emb [b, n, W] fp32; levels: per level (seg int64 [b, n] | None, lmask bool [b, S], S, width, p, given bool [b, S] | None) ->
    (emb with dropped columns zeroed, drop mask bool [b, n, W])."""
    require_gpu(emb)
    b, n, W = emb.shape
    emb = emb.contiguous()
    nl = len(levels)
    keep = []                                     # the contiguous / byte views must outlive the launch call

    def u8(t):
        t = _mask_u8(t)
        keep.append(t)
        return t
    segs = (ctypes.c_void_p * nl)(); lms = (ctypes.c_void_p * nl)(); givs = (ctypes.c_void_p * nl)()
    Ss = (ctypes.c_int * nl)(); cols = (ctypes.c_int * nl)(); ps = (ctypes.c_float * nl)()
    col = 0
    for i, (seg, lmask, S, width, p, given) in enumerate(levels):
        if seg is not None:
            seg = seg.contiguous(); keep.append(seg)
            if seg.dtype != torch.int64:
                raise SpnError("latent_drop: int64 segment ids")
        segs[i] = 0 if seg is None else seg.data_ptr()
        lms[i] = 0 if lmask is None else u8(lmask).data_ptr()
        givs[i] = 0 if given is None else u8(given).data_ptr()
        Ss[i], cols[i], ps[i] = int(S), col, float(p)
        col += int(width)
    if col != W:
        raise SpnError("latent_drop: the level widths must add up to the embedding width")
    out = torch.empty_like(emb)
    drop = torch.empty((b, n, W), device=emb.device, dtype=torch.bool)
    call("spn_latent_drop", c_int(nl), segs, lms, Ss, cols, ps, givs, c_int(int(bool(inclusive))), ptr(emb), ptr(u8(mask)),
         ptr(None if deadpan is None else u8(deadpan)), c_int(b), c_int(n), c_int(W), ctypes.c_uint(seed & 0xFFFFFFFF), ptr(out),
         ptr(drop.view(torch.uint8)), stream_ptr())
    return out, drop


def latent_drop_bwd(g: torch.Tensor, drop: torch.Tensor) -> torch.Tensor:
    g = g.contiguous()
    if g.dtype != F32:
        g = g.float()
    dx = torch.empty_like(g)
    call("spn_latent_drop_bwd", ptr(g), ptr(drop.view(torch.uint8)), c_long(g.numel()), ptr(dx), stream_ptr())
    return dx


# ---------------------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------------------

def sumsq(g: torch.Tensor, out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[0] += sum(g^2).  With `ws` (fp32 scratch of `sumsq_ws_floats()` elements): fixed summation order -- the same bits for the same
    g on every launch and every rank (what the data-parallel clip needs); without: block partials by float atomics."""
    if out is None:
        out = zeros_small(1, g.device)
    if ws is not None:
        call("spn_sumsq_det", ptr(g), c_long(g.numel()), ptr(out), ptr(ws), stream_ptr())
    else:
        call("spn_sumsq", ptr(g), c_long(g.numel()), ptr(out), stream_ptr())
    return out


def sumsq_ws_floats() -> int:
    return int(load().spn_sumsq_det_ws_floats())


def adamw_step(p, g, m, v, shadow, normsq, *, max_norm: float, grad_scale: float, lr: float, betas=(0.9, 0.999), eps=1e-8,
               weight_decay: float = 0.0, step: int = 1, slot_mask: Optional[torch.Tensor] = None):
    """`slot_mask`: uint8 [n / 8]; slots with 0 belong to parameters without a gradient this step and are left untouched."""
    if slot_mask is not None and (slot_mask.dtype != torch.uint8 or slot_mask.numel() * 8 != p.numel()):
        raise SpnError("adamw_step: slot_mask must be uint8 with one entry per 8 parameters")
    call("spn_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), ptr(slot_mask), c_long(p.numel()), ptr(normsq), c_float(max_norm or 0.0),
         c_float(grad_scale), c_float(lr), c_float(betas[0]), c_float(betas[1]), c_float(eps), c_float(weight_decay), c_int(step),
         stream_ptr())


# ---------------------------------------------------------------------------------------------------------
# opt-in launch timing (bench.py roofline leg): HIP events on the launch stream around GEMM / attention launches
# ---------------------------------------------------------------------------------------------------------

class _Profile:
    def __init__(self):
        self.enabled, self.records = False, []

    def enable(self):
        self.enabled, self.records = True, []

    def disable(self):
        self.enabled = False

    def wrap(self, name, flops, tag, fn):
        if not self.enabled:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((name, flops, s, e, tag))
        return out

    def collect(self):
        torch.cuda.synchronize()
        return [(n, f, s.elapsed_time(e), tag) for n, f, s, e, tag in self.records]


PROFILE = _Profile()
_gemm_raw, _attn_fwd_raw, _attn_bwd_raw = gemm, attn_fwd, attn_bwd


def gemm(a, b, *, ta=False, tb=False, **kw):  # noqa: F811
    if not PROFILE.enabled:
        return _gemm_raw(a, b, ta=ta, tb=tb, **kw)
    a2, b2 = _rows2d(a), _rows2d(b)
    M, K = (a2.shape[1], a2.shape[0]) if ta else (a2.shape[0], a2.shape[1])
    N = b2.shape[1] if tb else b2.shape[0]
    out = kw.get("out")
    f32 = (out.dtype if out is not None else kw.get("out_dtype", BF16)) == F32
    return PROFILE.wrap("gemm_bf16", 2.0 * M * N * K, f"{M}x{N}x{K}:{'T' if ta else 'N'}{'T' if tb else 'N'}:{'f32' if f32 else 'bf16'}",
                        lambda: _gemm_raw(a, b, ta=ta, tb=tb, **kw))


_gemm_glu_raw = gemm_glu


def gemm_glu(x, w, bias, **kw):  # noqa: F811
    if not PROFILE.enabled:
        return _gemm_glu_raw(x, w, bias, **kw)
    M, K = x.shape
    N = w.shape[0]
    return PROFILE.wrap("gemm_bf16", 2.0 * M * N * K, f"{M}x{N}x{K}:NN:bf16+glu", lambda: _gemm_glu_raw(x, w, bias, **kw))


_gemm_glu_bwd_raw = gemm_glu_bwd


def gemm_glu_bwd(dy, w2, u, **kw):  # noqa: F811
    if not PROFILE.enabled:
        return _gemm_glu_bwd_raw(dy, w2, u, **kw)
    M, K = dy.shape
    N = w2.shape[1]
    return PROFILE.wrap("gemm_bf16", 2.0 * M * N * K, f"{M}x{N}x{K}:NT:bf16+glu_bwd", lambda: _gemm_glu_bwd_raw(dy, w2, u, **kw))


def attn_fwd(q, k, v, **kw):  # noqa: F811
    if not PROFILE.enabled:
        return _attn_fwd_raw(q, k, v, **kw)
    b_, nq, h, dh = q.shape
    fl = 4.0 * b_ * h * nq * k.shape[1] * dh * (0.5 if kw.get("causal") else 1.0)
    return PROFILE.wrap("attn_fwd", fl, f"b{b_} h{h} nq{nq} nk{k.shape[1]} causal{int(bool(kw.get('causal')))}",
                        lambda: _attn_fwd_raw(q, k, v, **kw))


def attn_bwd(q, k, v, o, d_o, lse, **kw):  # noqa: F811
    if not PROFILE.enabled:
        return _attn_bwd_raw(q, k, v, o, d_o, lse, **kw)
    b_, nq, h, dh = q.shape
    fl = 10.0 * b_ * h * nq * k.shape[1] * dh * (0.5 if kw.get("causal") else 1.0)
    return PROFILE.wrap("attn_bwd", fl, f"b{b_} h{h} nq{nq} nk{k.shape[1]} causal{int(bool(kw.get('causal')))}",
                        lambda: _attn_bwd_raw(q, k, v, o, d_o, lse, **kw))


# HBM-bound kernels: the "work" of a record is ALGORITHMIC BYTES (every operand once), for the element-wise roofline object of bench.py
_ln_fwd_raw, _ln_bwd_raw, _act_bwd_raw, _act_fwd_raw = layernorm_fwd, layernorm_bwd, act_bwd, act_fwd


def _esz(t) -> int:
    return 0 if t is None else t.element_size()


def layernorm_fwd(x, gamma, beta, gb=None, out_dtype=BF16, eps=1e-5, out=None):  # noqa: F811
    if not PROFILE.enabled:
        return _ln_fwd_raw(x, gamma, beta, gb, out_dtype, eps, out)
    n = x.numel()
    ysz = out.element_size() if out is not None else (2 if out_dtype == BF16 else 4)
    by = n * (x.element_size() + ysz) + (2 * n * gb.element_size() if gb is not None else 0)
    return PROFILE.wrap("ln_fwd", float(by), f"T{n // x.shape[-1]} D{x.shape[-1]} {'ada' if gb is not None else 'ln'}",
                        lambda: _ln_fwd_raw(x, gamma, beta, gb, out_dtype, eps, out))


_adaln_fwd_raw = adaln_fwd


def adaln_fwd(x, cond, w, bias, eps=1e-5, want_gamma=True):  # noqa: F811
    if not PROFILE.enabled:
        return _adaln_fwd_raw(x, cond, w, bias, eps, want_gamma)
    n = x.numel()
    by = n * (4 + 2 + (2 if want_gamma else 0)) + cond.numel() * 2
    return PROFILE.wrap("ln_fwd", float(by), f"T{n // x.shape[-1]} D{x.shape[-1]} ada-fused", lambda: _adaln_fwd_raw(x, cond, w, bias, eps, want_gamma))


def layernorm_bwd(x, dy, gamma, gb, mean, rstd, *, dres=None, dx_dtype=F32, dgamma=None, dbeta=None, want_dgb=False, want_dx16=False,  # noqa: F811
                  dgb_colsum=None):
    run = lambda: _ln_bwd_raw(x, dy, gamma, gb, mean, rstd, dres=dres, dx_dtype=dx_dtype, dgamma=dgamma, dbeta=dbeta,   # noqa: E731
                              want_dgb=want_dgb, want_dx16=want_dx16, dgb_colsum=dgb_colsum)
    if not PROFILE.enabled:
        return run()
    n = x.numel()
    dxsz = 4 if dx_dtype == F32 else 2
    by = n * (x.element_size() + 2 + dxsz + (2 if (want_dx16 and dx_dtype == F32) else 0) + (4 if dres is not None else 0))
    by += (2 * n * gb.element_size() if gb is not None else 0) + (2 * n * 2 if want_dgb else 0)
    return PROFILE.wrap("ln_bwd", float(by), f"T{n // x.shape[-1]} D{x.shape[-1]} {'ada' if gb is not None else 'ln'}", run)


def act_bwd(u, dout, *, act, glu, p_drop=0.0, seed=0, colsum=None):  # noqa: F811
    run = lambda: _act_bwd_raw(u, dout, act=act, glu=glu, p_drop=p_drop, seed=seed, colsum=colsum)   # noqa: E731
    if not PROFILE.enabled:
        return run()
    return PROFILE.wrap("act_bwd", float(2 * u.numel() * 2 + dout.numel() * 2), f"T{u.numel() // u.shape[-1]} W{u.shape[-1]}", run)


def act_fwd(u, *, act, glu, p_drop=0.0, seed=0):  # noqa: F811
    run = lambda: _act_fwd_raw(u, act=act, glu=glu, p_drop=p_drop, seed=seed)   # noqa: E731
    if not PROFILE.enabled:
        return run()
    return PROFILE.wrap("act_fwd", float(u.numel() * 2 + (u.numel() // (2 if glu else 1)) * 2), f"T{u.numel() // u.shape[-1]} W{u.shape[-1]}", run)


# ---------------------------------------------------------------------------------------------------------
# decode (b = 1) kernels: fp32, position read from a device scalar so that one captured step can be replayed
# ---------------------------------------------------------------------------------------------------------

def dec_gemv(W, x, y, *, bias=None, residual=None, pos=None, x_ld=0, x_off=0, y_ld=0, y_off=0, kn_layout=False):
    N, K = (W.shape[1], W.shape[0]) if kn_layout else (W.shape[0], W.shape[1])
    call("spn_dec_gemv", ptr(W), c_long(W.stride(0)), ptr(x), c_long(x_ld), c_int(x_off), ptr(bias), ptr(residual), ptr(y),
         c_long(y_ld), c_int(y_off), ptr(pos), c_int(N), c_int(K), c_int(int(kn_layout)), stream_ptr())
    return y


def dec_embed(tables, tokens2d, pos, y, *, row_off=0, gamma=None, beta=None, eps=1e-5):
    E = [t.shape[1] for t in tables]
    call("spn_dec_embed", c_int(len(tables)), _ptr_array(tables), _int_array(E), ptr(tokens2d), c_long(tokens2d.stride(0)),
         c_int(row_off), ptr(pos), ptr(gamma), ptr(beta), ptr(y), c_float(eps), stream_ptr())
    return y


def dec_embed_proj(tables, tokens_a, tokens_b, pos, W, bias, y, *, gamma=None, beta=None, eps=1e-5):
    """y[0:N] / y[N:2N] = W . LN(tuple embedding of tokens_a[*pos] / tokens_b[*pos + 1]) + bias: dec_embed + dec_gemv for both sequences."""
    if tokens_a.stride(0) != tokens_b.stride(0):
        raise SpnError("dec_embed_proj: the two token arrays must share their row stride")
    E = [t.shape[1] for t in tables]
    call("spn_dec_embed_proj", c_int(len(tables)), _ptr_array(tables), _int_array(E), ptr(tokens_a), ptr(tokens_b), c_long(tokens_a.stride(0)),
         ptr(pos), ptr(gamma), ptr(beta), c_float(eps), ptr(W), c_long(W.stride(0)), ptr(bias), ptr(y), c_int(W.shape[0]), stream_ptr())
    return y


def dec_step_begin(tables, tokens_a, tokens_b, pos_next, pos, W, bias, y, *, gamma=None, beta=None, eps=1e-5, rider=None):
    """First launch of a fused decode step: `dec_embed_proj` reading the position from `pos_next`, republishing it in `pos` (the scalar
    every later launch of the step reads; the head launch writes position + 1 back into `pos_next`), and an optional rider GEMV
    `rider = (W_r, x2d, x_off, bias_r, y_r)`: y_r = W_r . x2d[*pos_next + x_off] + bias_r in the same launch."""
    if tokens_a.stride(0) != tokens_b.stride(0):
        raise SpnError("dec_step_begin: the two token arrays must share their row stride")
    E = [t.shape[1] for t in tables]
    if rider is not None:
        rW, rx, rx_off, rb, ry = rider
        rargs = (ptr(rW), c_long(rW.stride(0)), c_int(rW.shape[0]), c_int(rW.shape[1]), ptr(rx), c_long(rx.stride(0)), c_int(rx_off), ptr(rb), ptr(ry))
    else:
        rargs = (ptr(None), c_long(0), c_int(0), c_int(0), ptr(None), c_long(0), c_int(0), ptr(None), ptr(None))
    call("spn_dec_step_begin", c_int(len(tables)), _ptr_array(tables), _int_array(E), ptr(tokens_a), ptr(tokens_b), c_long(tokens_a.stride(0)),
         ptr(pos_next), ptr(pos), ptr(gamma), ptr(beta), c_float(eps), ptr(W), c_long(W.stride(0)), ptr(bias), ptr(y), c_int(W.shape[0]),
         *rargs, stream_ptr())
    return y


def dec_copy_row(src, dst, pos, D, *, src_ld=0, src_off=0, dst_ld=0, dst_off=0):
    call("spn_dec_copy_row", ptr(src), c_long(src_ld), c_int(src_off), ptr(dst), c_long(dst_ld), c_int(dst_off), ptr(pos), c_int(D),
         stream_ptr())


def dec_glu(u, out, I, *, act=0, glu=True):
    call("spn_dec_glu", ptr(u), ptr(out), c_int(I), c_int(act), c_int(int(glu)), stream_ptr())
    return out


def dec_attn(qkv, kcache, vcache, slopes, pos, o, *, h, kvh, scale):
    call("spn_dec_attn", ptr(qkv), ptr(kcache), ptr(vcache), ptr(slopes), ptr(pos), ptr(o), c_int(h), c_int(kvh), c_float(scale),
         stream_ptr())
    return o


def dec_argmax_write(logits, V, tokens2d, dim, pos, *, ban_mask=0b11, mask_id=1):
    call("spn_dec_argmax_write", ptr(logits), c_int(V), ctypes.c_uint(ban_mask), ptr(tokens2d), c_long(tokens2d.stride(0)), c_int(dim),
         c_int(mask_id), ptr(pos), stream_ptr())


def dec_fused_gemv(W, x, y, *, N=None, pos=None, x_ld=0, x_off=0, norm=0, gamma=None, beta=None, eps=1e-5, bias=None, residual=None,
                   y_ld=0, y_off=0, y2=None, y2_ld=0, y2_off=0, xn_out=None, xn_ld=0, xn_off=0, glu=0, act=-1):
    """y = [GLU](W . LN?(x) + bias) + residual in one launch (decode.hip).  norm: 0 none, 1 affine (gamma, beta), 2 adaptive
    (gamma = the (gamma|beta) row).  glu=1: W holds values|gates (2N rows); glu=-1 with act>=0: plain activation."""
    K = W.shape[1]
    N = N if N is not None else (W.shape[0] // 2 if glu == 1 else W.shape[0])
    call("spn_dec_fused_gemv", ptr(W), c_long(W.stride(0)), c_int(N), c_int(K), ptr(x), c_long(x_ld), c_int(x_off), c_int(norm),
         ptr(gamma), ptr(beta), c_float(eps), ptr(bias), ptr(residual), ptr(y), c_long(y_ld), c_int(y_off), ptr(y2), c_long(y2_ld),
         c_int(y2_off), ptr(xn_out), c_long(xn_ld), c_int(xn_off), c_int(glu), c_int(act), ptr(pos), stream_ptr())
    return y


def dec_cat(x, d, out, pos, *, gamma=None, beta=None, eps=1e-5, ctx=None, style=None):
    call("spn_dec_cat", ptr(x), c_int(d), ptr(gamma), ptr(beta), c_float(eps), ptr(ctx), c_long(ctx.stride(0) if ctx is not None else 0),
         c_int(ctx.shape[1] if ctx is not None else 0), ptr(style), c_long(style.stride(0) if style is not None else 0),
         c_int(style.shape[1] if style is not None else 0), ptr(pos), ptr(out), stream_ptr())
    return out


def dec_cat_gemv(W, x, d, y, pos, *, gamma=None, beta=None, eps=1e-5, ctx=None, style=None, bias=None, y2=None, y2_ld=0):
    """y = W . (LN?(x[:d]) | ctx[*pos + 1] | style[*pos + 1]) + bias, row *pos of y2 mirrored: `dec_cat` + `dec_fused_gemv` in one launch."""
    call("spn_dec_cat_gemv", ptr(W), c_long(W.stride(0)), c_int(W.shape[0]), ptr(x), c_int(d), ptr(gamma), ptr(beta), c_float(eps), ptr(ctx),
         c_long(ctx.stride(0) if ctx is not None else 0), c_int(ctx.shape[1] if ctx is not None else 0), ptr(style),
         c_long(style.stride(0) if style is not None else 0), c_int(style.shape[1] if style is not None else 0), ptr(bias), ptr(y), ptr(y2),
         c_long(y2_ld), ptr(pos), stream_ptr())
    return y


def dec_attn2(qkv, kcache, vcache, slopes, pos, o, part, counter, kmax2, *, h, kvh, scale, splits):
    call("spn_dec_attn2", ptr(qkv), ptr(kcache), ptr(vcache), ptr(slopes), ptr(pos), ptr(o), ptr(part), ptr(counter), ptr(kmax2),
         c_int(h), c_int(kvh), c_float(scale), c_int(splits), stream_ptr())
    return o


def dec_attn_out(W, part, y, *, h, splits, residual=None):
    """y = residual + W . (attention output merged from the partials left by dec_attn2 / dec_xattn with o=None)."""
    call("spn_dec_attn_out", ptr(W), c_long(W.stride(0)), c_int(W.shape[0]), ptr(part), c_int(h), c_int(splits), ptr(residual), ptr(y),
         stream_ptr())
    return y


def dec_xattn(q, kctx, vctx, slopes, kmask, o, part, counter, *, h, kvh, scale, splits, nk_dev=None):
    """Single-query cross-attention over the projected context rows (see spn_dec_xattn); kmask: uint8 [nk] or None.
    nk_dev: int32 device scalar with the number of valid context rows (kctx / vctx are then buffers of at least that many rows)."""
    if nk_dev is not None:
        call("spn_dec_xattn_dyn", ptr(q), ptr(kctx), ptr(vctx), ptr(slopes), ptr(kmask), ptr(nk_dev), ptr(o), ptr(part), ptr(counter),
             c_int(h), c_int(kvh), c_float(scale), c_int(splits), stream_ptr())
        return o
    call("spn_dec_xattn", ptr(q), ptr(kctx), ptr(vctx), ptr(slopes), ptr(kmask), c_int(kctx.shape[0]), ptr(o), ptr(part), ptr(counter),
         c_int(h), c_int(kvh), c_float(scale), c_int(splits), stream_ptr())
    return o


def dec_lookup(tab, pos, out):
    call("spn_dec_lookup", ptr(tab), ptr(pos), ptr(out), stream_ptr())


class DecPairArgs(ctypes.Structure):
    """include/spn.h: spn_dec_pair_args (field for field)."""
    _P, _L, _I, _F = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    _fields_ = [("Wqkv", _P), ("ld_qkv", _L), ("Wo", _P), ("ld_o", _L), ("W1", _P), ("ld_1", _L), ("b1", _P), ("W2", _P), ("ld_2", _L), ("b2", _P),
                ("slopes", _P), ("kcache", _P), ("vcache", _P), ("kmax2", _P), ("jlo", _P),
                ("norm1", _I), ("gam1", _P), ("bet1", _P), ("eps1", _F), ("norm2", _I), ("gam2", _P), ("bet2", _P), ("eps2", _F),
                ("x", _P), ("y2", _P), ("y2_ld", _L), ("d", _I), ("h", _I), ("kvh", _I), ("inner", _I), ("S", _I), ("act", _I), ("scale", _F),
                ("pos", _P), ("tick", _P), ("layer", _I), ("bump", _I), ("gq", _P), ("gp", _P), ("go", _P), ("gx", _P), ("gg", _P), ("gxo", _P), ("err", _P), ("stamps", _P)]


def dec_pair_groups(d, h, kvh, inner, S) -> int:
    """Workgroups of the persistent layer-pair launch for this shape; 0 = the shape is not supported (run the five launches)."""
    return int(load().spn_dec_pair_groups(int(d), int(h), int(kvh), int(inner), int(S)))


class DecChainExt(ctypes.Structure):
    """include/spn.h: spn_dec_chain_ext (field for field)."""
    _P, _L, _I, _F = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    _fields_ = [("Wm", _P), ("ld_m", _L), ("bm", _P), ("xin", _P), ("Km", _I), ("y2m", _P), ("y2m_ld", _L),
                ("Wp", _P), ("ld_p", _L), ("bp", _P), ("cat_gamma", _P), ("cat_beta", _P), ("cat_eps", _F),
                ("ctx", _P), ("ctx_ld", _L), ("ctx_w", _I), ("style", _P), ("style_ld", _L), ("style_w", _I), ("y2p", _P), ("y2p_ld", _L),
                ("gf", _P), ("gxf", _P), ("Wh", _P), ("ld_h", _L), ("Nh", _I), ("normh", _I), ("gamh", _P), ("beth", _P), ("epsh", _F),
                ("e_out", _P), ("xn_out", _P), ("xn_ld", _L),
                ("hn", _I), ("hD", _I), ("htable", _P * 16), ("hV", _I * 16), ("hwidth", _I * 16), ("hcol0", _I * 16), ("hdim", _I * 16),
                ("hgamma", _P), ("hbeta", _P), ("heps", _F), ("hban", ctypes.c_uint), ("tokens", _P), ("tok_ld", _L), ("mask_id", _I),
                ("pos_next", _P), ("ge", _P), ("gh", _P),
                ("en", _I), ("eD", _I), ("eN", _I), ("eR", _I), ("etable", _P * 16), ("ewidth", _I * 16), ("ecol0", _I * 16),
                ("tok_a", _P), ("tok_b", _P), ("etok_ld", _L), ("egamma", _P), ("ebeta", _P), ("eeps", _F), ("We", _P), ("ld_e", _L), ("be", _P),
                ("gin", _P),
                ("rW", _P), ("r_ldw", _L), ("rN", _I), ("rK", _I), ("rx", _P), ("rx_ld", _L), ("rx_rows", _I), ("rbias", _P), ("ry", _P),
                ("ada_par", _L), ("gt", _P), ("gl", _P), ("stopk", _P), ("sinv_temp", _F), ("sseed", _P)]


def _fill_struct(a, kw, keep, what):
    kw = dict(kw)
    for name, ctype in a._fields_:
        v = kw.pop(name, None)
        if ctype is ctypes.c_void_p:
            setattr(a, name, None if v is None else v.data_ptr())
            if v is not None:
                keep.append(v)
        elif isinstance(ctype, type) and issubclass(ctype, ctypes.Array):   # fixed-size arrays: a list of tensors (pointers) or of ints
            arr = getattr(a, name)
            vals = list(v) if v is not None else []
            if len(vals) > len(arr):
                raise SpnError(f"{what}: {name} takes at most {len(arr)} entries")
            for i in range(len(arr)):
                x = vals[i] if i < len(vals) else None
                if ctype._type_ is ctypes.c_void_p:
                    arr[i] = None if x is None else x.data_ptr()
                    if x is not None:
                        keep.append(x)
                else:
                    arr[i] = int(x) if x is not None else 0
        else:
            setattr(a, name, v if v is not None else 0)
    if kw:
        raise SpnError(f"{what}: unknown fields {sorted(kw)}")


class DecPairChain:
    """The argument records of a chain of decoder layer pairs (spn_dec_pairs): a host array and its device copy, built once per engine.
    `records`: one dict per pair with the fields of spn_dec_pair_args (tensors as tensors, None = null)."""

    def __init__(self, records, device, ext=None):
        for which, st in ((0, DecPairArgs), (1, DecChainExt)):
            if load().spn_dec_struct_size(which) != ctypes.sizeof(st):
                raise SpnError(f"{st.__name__}: record layout differs from include/spn.h ({ctypes.sizeof(st)} bytes here, "
                               f"{load().spn_dec_struct_size(which)} in the library)")
        self.n = len(records)
        self.host = (DecPairArgs * self.n)()
        self.keep = []                                   # the tensors behind the raw pointers
        for a, kw in zip(self.host, records):
            _fill_struct(a, kw, self.keep, "dec_pairs")
        self.dev = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(device)
        self.ext = self.ext_dev = None
        self.max_notes = 0                               # most notes any launch of this chain ran (tests)
        if ext:                                          # front / tail phases of the same launch: spn_dec_chain_ext
            self.ext = DecChainExt()
            self._ext_kw = dict(ext)
            _fill_struct(self.ext, ext, self.keep, "dec_pairs_ext")
            self.ext_dev = torch.frombuffer(bytearray(bytes(self.ext)), dtype=torch.uint8).to(device)

    def update_ext(self, **fields):
        """Change fields of the extension record (per-run operands: token buffer, tables) on the host AND in the device copy the launch
        reads; stream-ordered, so launches enqueued afterwards see the new record.  Not for use under graph capture."""
        cur = {name: getattr(self, "_ext_kw", {}).get(name) for name, _ in DecChainExt._fields_}
        cur.update(fields)
        self._ext_kw = {k: v for k, v in cur.items() if v is not None}
        keep = []
        _fill_struct(self.ext, self._ext_kw, keep, "dec_pairs_ext")
        self.keep_ext = keep
        self.ext_dev.copy_(torch.frombuffer(bytearray(bytes(self.ext)), dtype=torch.uint8))

    def launch(self, notes: int = 1):
        """`notes` > 1: that many consecutive notes in one launch (spn_dec_pairs_notes; needs the whole note in the launch and ext.gt)."""
        self.max_notes = max(self.max_notes, notes)
        if self.ext is not None and notes > 1:
            call("spn_dec_pairs_notes", self.host, ctypes.c_void_p(self.dev.data_ptr()), c_int(self.n), ctypes.byref(self.ext),
                 ctypes.c_void_p(self.ext_dev.data_ptr()), c_int(notes), stream_ptr())
        elif self.ext is not None:
            call("spn_dec_pairs_ext", self.host, ctypes.c_void_p(self.dev.data_ptr()), c_int(self.n), ctypes.byref(self.ext),
                 ctypes.c_void_p(self.ext_dev.data_ptr()), stream_ptr())
        else:
            call("spn_dec_pairs", self.host, ctypes.c_void_p(self.dev.data_ptr()), c_int(self.n), stream_ptr())


def dec_head(tables, col0, dims, D, e, gamma, beta, eps, tokens2d, pos, part, counter, *, slabs=8, ban_mask=0b11, mask_id=1, pos_next=None):
    """`pos_next`: int32 device scalar that receives position + 1 (the launch that closes a fused step, see `dec_step_begin`)."""
    V = [t.shape[0] for t in tables]
    W = [t.shape[1] for t in tables]
    call("spn_dec_head", c_int(len(tables)), _ptr_array(tables), _int_array(V), _int_array(W), _int_array(col0), _int_array(dims), c_int(D),
         ptr(e), ptr(gamma), ptr(beta), c_float(eps), ctypes.c_uint(ban_mask), ptr(tokens2d), c_long(tokens2d.stride(0)), c_int(mask_id),
         ptr(pos), ptr(part), ptr(counter), c_int(slabs), ptr(pos_next), stream_ptr())


def dec_attn_rows(q, kcache, vcache, slopes, t0, o, *, h, kvh, scale):
    """Causal single-query attention for rows t0 .. t0+n-1 at once (q: [n, >= h*64] fp32 view, o: [n, h*64])."""
    call("spn_dec_attn_rows", ptr(q), c_long(q.stride(0)), ptr(kcache), ptr(vcache), ptr(slopes), c_int(t0), c_int(q.shape[0]), ptr(o),
         c_long(o.stride(0)), c_int(h), c_int(kvh), c_float(scale), stream_ptr())
    return o


def dec_glu_rows(u, out, I, *, act=0, glu=True):
    call("spn_dec_glu_rows", ptr(u), c_long(u.stride(0)), ptr(out), c_long(out.stride(0)), c_int(u.shape[0]), c_int(I), c_int(act),
         c_int(int(glu)), stream_ptr())
    return out


def dec_head_sample(tables, col0, dims, D, e, gamma, beta, eps, tokens2d, pos, part, counter, logits, topk, seed, *, temperature=1.0,
                    slabs=8, ban_mask=0b11, mask_id=1, pos_next=None):
    """dec_head with top-k filtering + temperature + one multinomial draw per key instead of the arg-max (logits: fp32 scratch
    [n, ld]; topk: int32 [n] on the device; seed: int32/uint32 device scalar)."""
    V = [t.shape[0] for t in tables]
    W = [t.shape[1] for t in tables]
    call("spn_dec_head_sample", c_int(len(tables)), _ptr_array(tables), _int_array(V), _int_array(W), _int_array(col0), _int_array(dims),
         c_int(D), ptr(e), ptr(gamma), ptr(beta), c_float(eps), ctypes.c_uint(ban_mask), ptr(tokens2d), c_long(tokens2d.stride(0)),
         c_int(mask_id), ptr(pos), ptr(part), ptr(counter), c_int(slabs), ptr(logits), c_int(logits.stride(0)), ptr(topk),
         c_float(temperature), ptr(seed), ptr(pos_next), stream_ptr())


def dec_add_pos(pos, delta=1):
    call("spn_dec_add_pos", ptr(pos), c_int(delta), stream_ptr())


# ---------------------------------------------------------------------------------------------------------
# device-side batch builder (collate.hip)
# ---------------------------------------------------------------------------------------------------------

def collate_mixlm(score_flat, perf_flat, seg_flat, score_off, perf_off, deadpan, *, b, Ks, Kp, Ls, Lp, pad_id=0, mask_id=1,
                  label_pad_id=-100, ignore_ids=(), ignore_dims=0, label_pad_ignored_dims=True):
    """Raw ragged int32 device buffers -> dict of the padded tensors the reference's MixedLM collator returns."""
    require_gpu(score_flat, perf_flat, score_off, perf_off)
    for t in (score_flat, perf_flat, seg_flat, score_off, perf_off):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise SpnError("collate_mixlm: inputs must be contiguous int32")
    if deadpan is not None and deadpan.dtype not in (torch.uint8, torch.bool):
        raise SpnError("collate_mixlm: deadpan flags must be uint8/bool")
    dev = score_flat.device
    sum_s = score_flat.numel() // Ks
    i64 = dict(device=dev, dtype=torch.int64)
    out = {
        "score": torch.empty((b, Ls, Ks), **i64), "score_mask": torch.empty((b, Ls), device=dev, dtype=torch.bool),
        "score_len": torch.empty(b, **i64),
        "perf": torch.empty((b, Lp, Kp), **i64), "perf_mask": torch.empty((b, Lp), device=dev, dtype=torch.bool),
        "perf_len": torch.empty(b, **i64),
        "masked_perf": torch.empty((b, Lp, Kp), **i64), "labels": torch.empty((b, Lp, Kp), **i64),
        "deadpan_mask": torch.empty(b, device=dev, dtype=torch.bool),
    }
    if seg_flat is not None:
        for name in ("bar", "beat", "onset"):
            out[name] = torch.empty((b, Ls), **i64)
    ids = [int(t) for t in ignore_ids]
    call("spn_collate_mixlm", ptr(score_flat), ptr(perf_flat), ptr(seg_flat), ptr(score_off), ptr(perf_off), ptr(deadpan), c_int(b), c_int(Ks),
         c_int(Kp), c_int(Ls), c_int(Lp), c_long(sum_s), c_int(pad_id), c_int(mask_id), c_int(label_pad_id), _int_array(ids), c_int(len(ids)),
         ctypes.c_uint(ignore_dims), c_int(int(label_pad_ignored_dims)), ptr(out["score"]), ptr(out["score_mask"]), ptr(out["score_len"]),
         ptr(out["perf"]), ptr(out["perf_mask"]), ptr(out["perf_len"]), ptr(out["masked_perf"]), ptr(out["labels"]), ptr(out.get("bar")),
         ptr(out.get("beat")), ptr(out.get("onset")), ptr(out["deadpan_mask"]), stream_ptr())
    return out


def collate_pad_tokens(flat, off, *, b, K, L, pad_id=0):
    """Ragged int32 tokens [sum_n, K] + row offsets [b+1] -> (tokens int64 [b, L, K], mask bool [b, L], lengths int64 [b])."""
    require_gpu(flat, off)
    dev = flat.device
    out = torch.empty((b, L, K), device=dev, dtype=torch.int64)
    mask = torch.empty((b, L), device=dev, dtype=torch.bool)
    lens = torch.empty(b, device=dev, dtype=torch.int64)
    call("spn_collate_pad_tokens", ptr(flat), ptr(off), c_int(b), c_int(K), c_int(L), c_int(pad_id), ptr(out), ptr(mask), ptr(lens), stream_ptr())
    return out, mask, lens
