"""Transformer layer stack with the reference's contract (`modules/transformer/transformer.py:25-256`).

The residual stream is fp32; each sub-layer is  x <- x + f(norm(x))  with the add fused into the epilogue of f's last GEMM.
"""
import copy
from dataclasses import dataclass
from typing import List, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F_
from ...utils.config import DictConfig
from ..constructor import VariableModuleConfig, Constructor, Registry
from ..layers import Residual, AdaptiveLayerNorm, LayerNorm
from .attend import AttentionIntermediates
from .attention import Attention, AttentionConfig
from .feedforward import FeedForward, FeedForwardConfig
from ...utils.amp import no_autocast


@dataclass
class TransformerIntermediates:
    hiddens: Optional[List[Tensor]] = None
    attention: Optional[List[AttentionIntermediates]] = None


TransformerRegistry = type("_TransformerRegistry", (Registry,), {})()


@dataclass
class TransformerConfig(VariableModuleConfig):
    _target_: str = "default"
    dim: int = 512
    depth: int = 4
    heads: int = 8
    attention: Union[AttentionConfig, DictConfig] = None
    feed_forward: Union[FeedForwardConfig, DictConfig] = None
    causal: bool = False
    cross_attend: bool = False
    only_cross: bool = False
    pre_norm: bool = True
    use_adanorm: bool = False
    style_emb_dim: Optional[int] = None


@TransformerRegistry.register("default")
class Transformer(nn.Module, Constructor):
    # sub-layer pattern of one block, by (cross_attend, only_cross): self-attention 'a', cross-attention 'c', feed-forward 'f'
    _PATTERNS = {(False, False): "af", (False, True): "af", (True, False): "acf", (True, True): "cf"}

    def __init__(self, dim: int = 512, depth: int = 4, heads: int = 8, attention=None, feed_forward=None, causal: bool = False,
                 cross_attend: bool = False, only_cross: bool = False, pre_norm: bool = True, use_adanorm: bool = False,
                 style_emb_dim: Optional[int] = None):
        super().__init__()
        if use_adanorm and style_emb_dim is None:
            raise AssertionError('condition_dim should be provided with adanorm')
        attn_cfg = attention or AttentionConfig()
        ffn_cfg = feed_forward or FeedForwardConfig()
        self.dim, self.depth, self.pre_norm, self.ada_norm, self.cross_attend = dim, depth, pre_norm, use_adanorm, cross_attend

        def new_norm():   # one normalisation module of this stack (adaptive: conditioned on the style embedding)
            return AdaptiveLayerNorm(dim, style_emb_dim) if use_adanorm else LayerNorm(dim)

        make = {   # constructors of the three sub-layer kinds
            "a": lambda: Attention.init(config=attn_cfg, dim=dim, heads=heads, causal=causal),
            "c": lambda: Attention.init(config=attn_cfg, dim=dim, heads=heads),
            "f": lambda: FeedForward.init(config=ffn_cfg, dim=dim),
        }
        self.layer_types = tuple(self._PATTERNS[(bool(cross_attend), bool(only_cross))]) * depth
        self.num_attn_layers = self.layer_types.count("a")
        # registration order = state_dict order of the reference: `layers.<i>.0.<slot>` norms, `.1` the sub-layer, `.2` the residual, then
        # `final_norm`; slot 0 is the pre-norm, slot 2 the post-norm, slot 1 stays empty
        self.layers = nn.ModuleList()
        self.final_norm = new_norm() if pre_norm else nn.Identity()
        for kind in self.layer_types:
            block = make[kind]()
            slots = [new_norm() if pre_norm else None, None, None if pre_norm else new_norm()]
            self.layers.append(nn.ModuleList([nn.ModuleList(slots), block, Residual(dim)]))

    def _norm(self, norm, x, style, out_fp32=False, fork=False):
        if self.ada_norm:
            return norm(x, condition=style, out_fp32=out_fp32, fork=fork)
        return norm(x, out_fp32=out_fp32, fork=fork)

    @no_autocast
    def forward(self, x: Tensor, mask: Optional[Tensor] = None, context: Optional[Tensor] = None,
                context_mask: Optional[Tensor] = None, attn_mask: Optional[Tensor] = None,
                style_embeddings: Optional[Tensor] = None, mems: Optional[List[Tensor]] = None,
                intermediates_cache: Optional[TransformerIntermediates] = None, return_hiddens: bool = False):
        assert not (self.cross_attend ^ (context is not None)), 'context must be passed in if cross_attend is set to True'
        assert not self.ada_norm or style_embeddings is not None, 'style_embeddings must be passed for AdaLayerNorm'
        if mems is not None or attn_mask is not None:
            raise NotImplementedError("mems / attn_mask are not reachable from the ScorePerformer models")
        hiddens, attn_intermediates = [], []
        has_cache = intermediates_cache is not None
        intermediates_cache = copy.copy(intermediates_cache) if has_cache else None
        if has_cache:
            intermediates_cache.hiddens = list(intermediates_cache.hiddens)
            intermediates_cache.attention = list(intermediates_cache.attention)
        x = x[:, -1:] if has_cache else x
        if has_cache and style_embeddings is not None:
            style_embeddings = style_embeddings[:, -1:]
        if self.ada_norm and style_embeddings is not None:
            style_embeddings = F_.share_cond(style_embeddings)   # the norms' condition gradients accumulate in one buffer
        x = F_.cast(x, torch.float32)  # fp32 residual stream
        ctx_b = F_.cast(context, torch.bfloat16) if context is not None else None

        for layer_type, (norm, block, residual_fn) in zip(self.layer_types, self.layers):
            cache = None
            if layer_type == 'a':
                if has_cache:
                    cache_h = intermediates_cache.hiddens.pop(0)
                    x = torch.cat([cache_h, x], dim=1)
                if return_hiddens:
                    hiddens.append(x)
                x = x[:, -1:] if has_cache else x
            if has_cache and layer_type in ('a', 'c'):
                cache = intermediates_cache.attention.pop(0)
            residual = x
            pre_norm, post_branch_norm, post_main_norm = norm
            if pre_norm is not None and x.dtype == torch.float32 and x.requires_grad:
                # (norm(x), x) from one autograd node: its backward adds the two branch gradients inside the LayerNorm kernel
                h, residual = self._norm(pre_norm, x, style_embeddings, fork=True)
            else:
                h = self._norm(pre_norm, x, style_embeddings) if pre_norm is not None else F_.cast(x, torch.bfloat16)
            fuse = residual_fn.is_plain
            res_arg = residual if fuse else None
            if layer_type == 'a':
                out, inter, _ = block(h, mask=mask, cache=cache, residual=res_arg)
            elif layer_type == 'c':
                # under the cache protocol the block sees ONE query: its row mask is the last column.  (The reference passes the
                # whole-prefix mask here, transformer.py:201, and attention.py:216-218 then BROADCASTS the single output row to the
                # prefix length -- the rows pile up in its caches and `hidden_state[:, idx - 1]` reads a stale position from the third
                # decoded note on; oracle/ref_cpu.py greedy_unmask reproduces that defect to pin itself, the product does not.)
                qmask = mask[:, -1:] if (has_cache and mask is not None) else mask
                out, inter, _ = block(h, context=ctx_b, mask=qmask, context_mask=context_mask, residual=res_arg)
            else:
                out = block(h, residual=res_arg)
            x = out if fuse else residual_fn(out, residual)
            if return_hiddens and layer_type in ('a', 'c'):
                attn_intermediates.append(inter)
            if post_main_norm is not None:
                x = self._norm(post_main_norm, x, style_embeddings, out_fp32=True)

        if not isinstance(self.final_norm, nn.Identity):
            x = self._norm(self.final_norm, x, style_embeddings, out_fp32=True)
        if has_cache:
            cache_h = intermediates_cache.hiddens.pop(0)
            x = torch.cat([cache_h, x], dim=1)
        if return_hiddens:
            hiddens.append(x)
            return x, TransformerIntermediates(hiddens=hiddens, attention=attn_intermediates)
        return x


@dataclass
class EncoderConfig(TransformerConfig):
    _target_: str = "encoder"
    causal: bool = False


@TransformerRegistry.register("encoder")
class Encoder(Transformer):
    def __init__(self, **kwargs):
        super().__init__(causal=False, **kwargs)


@dataclass
class DecoderConfig(TransformerConfig):
    _target_: str = "decoder"
    causal: bool = True


@TransformerRegistry.register("decoder")
class Decoder(Transformer):
    def __init__(self, **kwargs):
        super().__init__(causal=True, **kwargs)
