"""Transformer attention layer with the reference's contract (`modules/transformer/attention.py:22-222`).

HIP execution plan per call: one fused QKV projection GEMM (q | k | v columns of one buffer; MQA = 512+64+64 columns),
the flash-style attention kernel reading that buffer in place through strides, and the out-projection GEMM whose
epilogue applies the query-row mask (attention.py:216-218) and, inside a Transformer stack, the residual add.

Options of the reference class beyond the shipped recipes:
* `dim_head` < 64: the kernels are built for 64-wide heads; narrower heads run through them with the head dimension zero-padded ON THE
  WEIGHTS (q / k / v rows and out-projection columns of the padding are zero, so scores and outputs are exactly those of the narrow
  head; `scale` stays dim_head ** -0.5).  Parameters and state_dict keep the reference's shapes.  `dim_head` > 64 raises.
* `num_mem_kv` > 0 (attention.py:98-101,146-153): learned memory keys / values per head in front of the sequence's keys, always
  visible (the key mask is padded with True, causal masks start behind them), positions shifted by their number in the ALiBi bias
  exactly as `rel_pos.get_bias(i, j, k=j - i)` shifts them.  Multi-head K/V only: with `one_kv_head` the reference itself fails in
  `torch.cat((mem_k, k))` ([b, h, m, d] against [b, n, d]).
* `max_attend` (attention.py:170-187) raises: the reference's mask `(-max_attend < dist) | (dist > max_attend)` is true for every
  key except those at least `max_attend` positions in a query's FUTURE -- with `causal` every score is masked -- so there is no
  behaviour worth reproducing; no recipe sets it.
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F_
from ...utils import default
from ..constructor import Constructor, ModuleConfig
from .attend import AttentionIntermediates, Attend
from .embeddings import ALiBiPositionalBias, LearnedALiBiPositionalBias
from ...utils.amp import no_autocast


@dataclass
class AttentionSharedIntermediates:
    rel_pos_bias: Optional[Tensor] = None


@dataclass
class AttentionConfig(ModuleConfig):
    dim: int = 512
    dim_head: int = 64
    heads: int = 8
    causal: bool = False
    dropout: float = 0.
    one_kv_head: bool = False
    num_mem_kv: int = 0
    shared_kv: bool = False
    value_dim_head: Optional[int] = None
    max_attend_past: Optional[int] = None
    alibi_pos_bias: bool = False
    alibi_num_heads: Optional[int] = None
    alibi_symmetric: bool = True
    alibi_learned: bool = False


class Attention(nn.Module, Constructor):
    # parameter groups that the ParamArena lays out contiguously and exposes as one fused GEMM operand
    _spn_fuse_groups = {"_w_qkv": ("to_q.weight", "to_k.weight", "to_v.weight"), "_w_kv": ("to_k.weight", "to_v.weight")}

    def __init__(self, dim: int, dim_head: int = 64, heads: int = 8, causal: bool = False, dropout: float = 0.,
                 one_kv_head: bool = False, num_mem_kv: int = 0, max_attend: Optional[int] = None, alibi_pos_bias: bool = False,
                 alibi_num_heads: Optional[int] = None, alibi_symmetric: bool = True, alibi_learned: bool = False):
        super().__init__()
        if dim_head > 64:
            raise NotImplementedError("the attention kernels are built for head widths up to 64 (every shipped recipe uses 64)")
        if max_attend is not None:
            raise NotImplementedError("max_attend: the reference's window mask hides every key but the far future (see the module "
                                      "docstring); not implemented")
        if num_mem_kv > 0 and one_kv_head:
            raise NotImplementedError("num_mem_kv with one_kv_head: the reference concatenates [b, h, m, d] memories with [b, n, d] "
                                      "keys and fails; not defined")
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.heads, self.causal, self.max_attend = heads, causal, max_attend
        self.one_kv_head = one_kv_head
        self.kv_heads = 1 if one_kv_head else heads
        out_dim = q_dim = dim_head * heads
        kv_dim = dim_head if one_kv_head else dim_head * heads
        self.to_q = nn.Linear(dim, q_dim, bias=False)
        self.to_k = nn.Linear(dim, kv_dim, bias=False)
        self.to_v = nn.Linear(dim, kv_dim, bias=False)
        self.rel_pos = None
        if alibi_pos_bias:
            alibi_num_heads = default(alibi_num_heads, heads)
            assert alibi_num_heads <= heads, 'number of ALiBi heads must be less than the total number of heads'
            klass = LearnedALiBiPositionalBias if alibi_learned else ALiBiPositionalBias
            self.rel_pos = klass(heads=alibi_num_heads, total_heads=heads, symmetric=alibi_symmetric or causal)
        self.attend = Attend(causal=causal, dropout=dropout, scale=self.scale)
        self.num_mem_kv = num_mem_kv
        if num_mem_kv > 0:    # attention.py:98-101
            self.mem_k = nn.Parameter(torch.randn(heads, num_mem_kv, dim_head))
            self.mem_v = nn.Parameter(torch.randn(heads, num_mem_kv, dim_head))
        self.to_out = nn.Linear(out_dim, dim, bias=False)
        self._w_qkv = None   # fused arena views (set by ParamArena)
        self._w_kv = None

    def _fused(self, name, parts):
        w = getattr(self, name)
        if self.dim_head == 64 and w is not None and w.device == parts[0].device:
            return w
        return torch.cat([self._head_rows(p) for p in parts], dim=0)

    def _head_rows(self, w: Tensor) -> Tensor:
        """[heads * dim_head, dim] -> [heads * 64, dim]: every head's rows followed by zero rows (dim_head < 64)."""
        if self.dim_head == 64:
            return w
        hh = w.shape[0] // self.dim_head
        return nn.functional.pad(w.view(hh, self.dim_head, -1), (0, 0, 0, 64 - self.dim_head)).reshape(hh * 64, -1)

    def _head_cols(self, w: Tensor) -> Tensor:
        """to_out.weight [dim, heads * dim_head] -> [dim, heads * 64] with zero columns behind every head's."""
        if self.dim_head == 64:
            return w
        return nn.functional.pad(w.view(w.shape[0], self.heads, self.dim_head), (0, 64 - self.dim_head)).reshape(w.shape[0], self.heads * 64)

    def _cache_rows(self, t: Tensor) -> Tensor:
        """Cached keys / values as this module hands them out ([b, n, dim_head] multi-query, [b, h, n, dim_head] otherwise) -> the kernels'
        row layout [b, n, kv_heads * 64]: narrow heads get their zero columns back (the intermediates are sliced to dim_head)."""
        if self.dim_head != 64:
            t = nn.functional.pad(t, (0, 64 - self.dim_head))
        return t if t.ndim == 3 else t.permute(0, 2, 1, 3).flatten(-2)

    def _memory_rows(self, b: int) -> Tensor:
        """[b, m, 2 * heads * 64] bf16: the learned memories in the fused (k | v) row layout of the sequence's own projections."""
        def rows(t):   # [h, m, d] -> [m, h * 64]
            t = nn.functional.pad(t, (0, 64 - self.dim_head)) if self.dim_head != 64 else t
            return t.permute(1, 0, 2).reshape(self.num_mem_kv, self.heads * 64)
        mem = F_.cast(torch.cat([rows(self.mem_k), rows(self.mem_v)], dim=-1), torch.bfloat16)
        return mem[None].expand(b, -1, -1)

    @no_autocast
    def forward(self, x: Tensor, context: Optional[Tensor] = None, mask: Optional[Tensor] = None,
                context_mask: Optional[Tensor] = None, attn_mask: Optional[Tensor] = None, prev_attn: Optional[Tensor] = None,
                mem: Optional[Tensor] = None, cache: Optional[AttentionIntermediates] = None,
                shared_cache: Optional[AttentionSharedIntermediates] = None, residual: Optional[Tensor] = None):
        b, n = x.shape[:2]
        h, kvh = self.heads, self.kv_heads
        has_context, has_mem, has_cache = context is not None, mem is not None, cache is not None
        assert not (has_mem and has_cache), 'cache is not compatible with memory keys'
        assert not (has_context and has_cache), 'cache is not compatible with context yet'
        if has_mem or attn_mask is not None or prev_attn is not None:
            raise NotImplementedError("mem / attn_mask / prev_attn are not reachable from the ScorePerformer models")
        p_drop = self.attend.dropout if self.training else 0.0
        slopes = self.rel_pos.padded_slopes() if self.rel_pos is not None else None
        key_mask = mask if context_mask is None else context_mask
        has_memkv = self.num_mem_kv > 0
        if has_memkv and key_mask is not None:     # attention.py:151-152: the memories are always visible
            key_mask = nn.functional.pad(key_mask, (self.num_mem_kv, 0), value=True)

        # the rows this block's output is multiplied by at the end (attention.py:216-218): handed to the attention core as well, which
        # then neither computes nor differentiates the padding rows of a ragged batch
        qmask = mask
        if mask is not None and has_cache:
            qmask = mask[:, -1:]
        core_qmask = qmask if qmask is not None and qmask.shape[1] == n else None

        if not has_context and not has_cache and not has_memkv:
            qkv = F_.linear(x, self._fused("_w_qkv", (self.to_q.weight, self.to_k.weight, self.to_v.weight)))
            o = F_.SelfAttnFn.apply(qkv, slopes, key_mask, h, kvh, self.causal, self.scale, p_drop, core_qmask)
            k_view, v_view = qkv[..., h * 64:(h + kvh) * 64], qkv[..., (h + kvh) * 64:]
        else:
            q = F_.linear(x, self._head_rows(self.to_q.weight))
            kv = F_.linear(context if has_context else x, self._fused("_w_kv", (self.to_k.weight, self.to_v.weight)))
            if has_memkv:  # attention.py:146-150: memories in front of the sequence's own keys / values (and behind an older cache)
                kv = torch.cat([self._memory_rows(b), kv], dim=1)
            if has_cache:  # attention.py:155-156 (K/V of earlier positions; layout b n (kvh d))
                ck, cv = self._cache_rows(cache.keys), self._cache_rows(cache.values)
                kv = torch.cat([torch.cat([ck, kv[..., :kvh * 64]], dim=1), torch.cat([cv, kv[..., kvh * 64:]], dim=1)], dim=-1)
            o = F_.CrossAttnFn.apply(q, kv, slopes, key_mask, h, kvh, self.causal, self.scale, p_drop, core_qmask)
            k_view, v_view = kv[..., :kvh * 64], kv[..., kvh * 64:]

        out = F_.linear(o, self._head_cols(self.to_out.weight), residual=residual, rowmask=qmask.contiguous() if qmask is not None else None,
                        out_fp32=residual is not None)
        if kvh != 1:  # reference layout b h n d
            k_view = k_view.unflatten(-1, (kvh, 64)).permute(0, 2, 1, 3)[..., :self.dim_head]
            v_view = v_view.unflatten(-1, (kvh, 64)).permute(0, 2, 1, 3)[..., :self.dim_head]
        elif self.dim_head != 64:
            k_view, v_view = k_view[..., :self.dim_head], v_view[..., :self.dim_head]
        return out, AttentionIntermediates(keys=k_view, values=v_view), AttentionSharedIntermediates(rel_pos_bias=None)
