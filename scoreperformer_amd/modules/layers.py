"""`Residual` and `AdaptiveLayerNorm` with the reference's constructor/forward contract (`modules/layers.py:13-47`),
executed by the HIP LayerNorm kernels."""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F_
from ..utils.amp import no_autocast


class Residual(nn.Module):
    def __init__(self, dim: int, scale_residual: bool = False, scale_residual_constant: float = 1.):
        super().__init__()
        self.residual_scale = nn.Parameter(torch.ones(dim)) if scale_residual else None
        self.scale_residual_constant = scale_residual_constant

    @no_autocast
    def forward(self, x, residual):
        # the transformer stack fuses the plain residual add into the producing GEMM's epilogue; this eager path only
        # serves stand-alone use and the (unused in shipped recipes) scaled variants
        skip = residual
        for factor in (self.residual_scale, None if self.scale_residual_constant == 1 else self.scale_residual_constant):
            if factor is not None:
                skip = skip * factor
        return skip + x.to(skip.dtype)

    @property
    def is_plain(self):
        return self.residual_scale is None and self.scale_residual_constant == 1


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters (same state_dict keys), HIP forward/backward; returns bf16 unless `out_fp32`."""

    @no_autocast
    def forward(self, x: Tensor, out_fp32: bool = False, fork: bool = False):
        """`fork=True` returns (norm(x), x): the pair a pre-norm residual block consumes, with one fused backward."""
        return F_.layer_norm(x, self.weight, self.bias, out_fp32=out_fp32, eps=self.eps, fork=fork)


class AdaptiveLayerNorm(nn.Module):
    def __init__(self, dim: int, condition_dim: int, eps: float = 1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        # parameter-free norm + Linear(condition -> gamma | beta); the bias starts at (1 | 0), i.e. a plain normalisation
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.linear = nn.Linear(condition_dim, 2 * dim)
        with torch.no_grad():
            self.linear.bias.copy_(torch.cat([torch.ones(dim), torch.zeros(dim)]))

    @no_autocast
    def forward(self, x: Tensor, condition: Optional[Tensor] = None, out_fp32: bool = False, fork: bool = False):
        if condition is None:  # gamma = 1, beta = 0
            return F_.layer_norm(x, None, None, out_fp32=out_fp32, eps=self.eps, fork=fork)
        if condition.ndim == 2:
            condition = condition.unsqueeze(1)
        if condition.shape[1] != x.shape[1]:
            condition = condition.expand(-1, x.shape[1], -1)
        return F_.ada_layer_norm(x, condition, self.linear.weight, self.linear.bias, out_fp32=out_fp32, eps=self.eps, fork=fork)
