"""Transformer feed-forward (`modules/transformer/feedforward.py:13-64`): GEMM(+bias) -> GLU*SiLU / GELU -> GEMM(+residual)."""
from dataclasses import dataclass

import torch.nn as nn

from ... import functional as F_
from ..constructor import Constructor, ModuleConfig
from ...utils.amp import no_autocast


class GLU(nn.Module):
    def __init__(self, dim_in, dim_out, activation):
        super().__init__()
        self.act = activation
        self.proj = nn.Linear(dim_in, dim_out * 2)

    @no_autocast
    def forward(self, x, p_drop: float = 0.0):
        return F_.linear_glu(x, self.proj.weight, self.proj.bias, act=F_.ACT_SILU if isinstance(self.act, nn.SiLU) else F_.ACT_GELU,
                            p_drop=p_drop)


@dataclass
class FeedForwardConfig(ModuleConfig):
    dim: int = 512
    mult: int = 4
    glu: bool = False
    swish: bool = False
    post_act_ln: bool = False
    dropout: float = 0.
    no_bias: bool = True


class FeedForward(nn.Module, Constructor):
    def __init__(self, dim: int = 512, mult: int = 4, glu: bool = False, swish: bool = False, post_act_ln: bool = False,
                 dropout: float = 0., no_bias: bool = True):
        super().__init__()
        inner_dim = int(dim * mult)
        activation = nn.SiLU() if swish else nn.GELU()
        project_in = nn.Sequential(nn.Linear(dim, inner_dim, bias=not no_bias), activation) if not glu \
            else GLU(dim, inner_dim, activation)
        self.ff = nn.Sequential(project_in, nn.LayerNorm(inner_dim) if post_act_ln else nn.Identity(), nn.Dropout(dropout),
                                nn.Linear(inner_dim, dim, bias=not no_bias))
        self.glu, self.act_code, self.dropout = glu, (F_.ACT_SILU if swish else F_.ACT_GELU), dropout

    @no_autocast
    def forward(self, x, residual=None):
        p = self.dropout if self.training else 0.0
        has_ln = isinstance(self.ff[1], nn.LayerNorm)
        out = self.ff[3]
        if self.glu and not has_ln:   # one autograd node for the block (activation backward inside the output projection's dX GEMM)
            proj = self.ff[0].proj
            return F_.feed_forward_glu(x, proj.weight, proj.bias, out.weight, out.bias, residual=residual, act=self.act_code, p_drop=p)
        p_act = 0.0 if has_ln else p     # with a post-activation LayerNorm the Dropout sits BEHIND the norm (feedforward.py:56-59)
        if self.glu:
            g = self.ff[0](x, p_drop=p_act)
        else:
            lin = self.ff[0][0]
            g = F_.glu_act(F_.linear(x, lin.weight, lin.bias), act=self.act_code, glu=False, p_drop=p_act, bias=lin.bias)
        if has_ln:
            g = F_.dropout(F_.layer_norm(g, self.ff[1].weight, self.ff[1].bias, eps=self.ff[1].eps), p, self.training)
        return F_.linear(g, out.weight, out.bias, residual=residual, out_fp32=residual is not None)
