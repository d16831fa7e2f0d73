"""Attention core (`modules/transformer/attend.py:27-186`) on the fused HIP attention kernels."""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn, Tensor

from ... import functional as F_
from ...utils.amp import no_autocast


@dataclass
class AttentionIntermediates:
    keys: Optional[Tensor] = None
    values: Optional[Tensor] = None
    qk_similarities: Optional[Tensor] = None

    def to_tuple(self):
        return self.keys, self.values, self.qk_similarities


class Attend(nn.Module):
    """softmax(q k^T * scale + slope_h * -|j - (i + nk - nq)| + key/causal mask) v without materialising the scores.

    `forward` accepts the reference layout (q: b h n d; k, v: b n d for multi-query or b h n d) with the masks the
    reference's `Attention` can actually produce: a key-padding mask [b, j] (or [b,1,1,j]) and causal masking;
    ALiBi is passed as per-head `slopes` instead of a materialised (h,i,j) bias.
    """

    def __init__(self, *, dropout: float = 0., causal: bool = False, scale: Optional[float] = None):
        super().__init__()
        self.scale = scale
        self.causal = causal
        self.dropout = dropout
        self.attn_dropout = nn.Dropout(dropout)
        self.efficient = True

    @no_autocast
    def forward(self, q, k, v, mask=None, attn_bias=None, prev_attn=None, slopes: Optional[Tensor] = None):
        assert prev_attn is None, 'residual attention not compatible with efficient attention'
        if attn_bias is not None:
            raise NotImplementedError("pass ALiBi as per-head `slopes`; a materialised attn_bias is never built here")
        b, h, n, d = q.shape
        if mask is not None and mask.ndim == 4:
            if mask.shape[1] != 1 or mask.shape[2] != 1:
                raise NotImplementedError("only key-padding masks (b,1,1,j) and causal masking are supported")
            mask = mask[:, 0, 0]
        qb = F_.cast(q, torch.bfloat16).permute(0, 2, 1, 3)                       # b n h d view
        kb = F_.cast(k, torch.bfloat16)
        vb = F_.cast(v, torch.bfloat16)
        kb = kb.unsqueeze(2) if kb.ndim == 3 else kb.permute(0, 2, 1, 3)
        vb = vb.unsqueeze(2) if vb.ndim == 3 else vb.permute(0, 2, 1, 3)
        kv = torch.cat([kb, vb], dim=-2).flatten(-2)               # b j (2*kvh*d)   (stand-alone path only)
        out = F_.CrossAttnFn.apply(qb.reshape(b, n, h * d), kv, slopes, mask, h, kb.shape[2], self.causal,
                                   self.scale if self.scale is not None else d ** -0.5, self.dropout if self.training else 0.0)
        return out.view(b, n, h, d).permute(0, 2, 1, 3), AttentionIntermediates(keys=k, values=v)
