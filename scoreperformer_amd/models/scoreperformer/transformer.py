"""TupleTransformer with the reference's contract (`models/scoreperformer/transformer.py:24-222`)."""
from dataclasses import dataclass, field
from typing import Optional, Union, Dict, List

import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F_
from ...modules.constructor import Constructor, ModuleConfig
from ...modules.layers import LayerNorm
from ...modules.transformer import (
    TransformerConfig, TransformerRegistry, TransformerIntermediates, AbsolutePositionalEmbedding
)
from ...utils import ExplicitEnum
from ...utils.config import DictConfig, MISSING
from .embeddings import (
    TupleTokenEmbeddingsConfig, TupleTokenEmbeddingsRegistry, TupleTokenHeadsConfig, TupleTokenHeadsRegistry,
    TupleTokenRegressionHeadConfig, TupleTokenRegressionHead
)
from ...utils.amp import no_autocast


class EmbeddingModes(ExplicitEnum):
    SUM = "mean"
    CONCAT = "cat"
    ATTENTION = "attention"
    ADANORM = "adanorm"


@dataclass
class TupleTransformerCaches:
    token_emb: Optional[Tensor] = None
    transformer: Optional[TransformerIntermediates] = None


@dataclass
class TupleTransformerOutput:
    hidden_state: Tensor
    logits: Optional[Dict[str, Tensor]] = None
    attentions: Optional[List[Tensor]] = None
    caches: Optional[TupleTransformerCaches] = None
    reg_values: Optional[Dict[str, Tensor]] = None
    # extensions (not in the reference): fused cross-entropy sums / argmax per key when `labels` is passed down
    ce_sums: Optional[Dict[str, Tensor]] = None
    argmax: Optional[Dict[str, Tensor]] = None
    eval_sums: Optional[Dict[str, Tensor]] = None   # key -> (#correct, distance sum) when an evaluator is attached (section 8(f) N3)


@dataclass
class TupleTransformerConfig(ModuleConfig):
    num_tokens: Dict[str, int] = MISSING
    dim: int = 512
    max_seq_len: int = 1024
    transformer: Union[DictConfig, TransformerConfig] = field(default_factory=lambda: TransformerConfig(_target_="default"))
    token_embeddings: Union[DictConfig, TupleTokenEmbeddingsConfig] = field(default_factory=TupleTokenEmbeddingsConfig)
    use_abs_pos_emb: bool = True
    emb_norm: bool = False
    emb_dropout: float = 0.0
    context_emb_dim: Optional[int] = None
    context_emb_mode: str = EmbeddingModes.ATTENTION
    style_emb_dim: Optional[int] = None
    style_emb_mode: str = EmbeddingModes.CONCAT
    lm_head: Optional[Union[DictConfig, TupleTokenHeadsConfig]] = None
    regression_head: Optional[Union[DictConfig, TupleTokenRegressionHeadConfig]] = None


class TupleTransformer(nn.Module, Constructor):
    def __init__(self, num_tokens: Dict[str, int], dim: int = 512, max_seq_len: int = 1024, transformer=None,
                 token_embeddings=None, use_abs_pos_emb: bool = True, emb_norm: bool = False, emb_dropout: float = 0.0,
                 context_emb_dim: Optional[int] = None, context_emb_mode: str = EmbeddingModes.ATTENTION,
                 style_emb_dim: Optional[int] = None, style_emb_mode: str = EmbeddingModes.CONCAT, lm_head=None,
                 regression_head=None):
        super().__init__()
        stack_cfg = transformer if transformer is not None else TransformerConfig(_target_="default")
        emb_cfg = token_embeddings if token_embeddings is not None else TupleTokenEmbeddingsConfig()
        self.dim, self.max_seq_len = dim, max_seq_len
        self.context_emb_dim, self.context_emb_mode = (context_emb_dim or 0), context_emb_mode
        self.style_emb_dim, self.style_emb_mode = (style_emb_dim or 0), style_emb_mode
        cat_context, cat_style = context_emb_mode == EmbeddingModes.CONCAT, style_emb_mode == EmbeddingModes.CONCAT

        # sub-modules in the reference's registration order (= its state_dict order): token_emb, transformer, pos_emb, emb_norm,
        # emb_dropout, project_emb, lm_head, regression_head
        self.token_emb = TupleTokenEmbeddingsRegistry.instantiate(
            config=emb_cfg, num_tokens=num_tokens, emb_dims=emb_cfg.get("emb_dims", dim), project_emb_dim=dim)
        if context_emb_mode != EmbeddingModes.ATTENTION:
            stack_cfg.cross_attend = False        # a context that is concatenated / summed is not cross-attended to
        self.transformer = TransformerRegistry.instantiate(
            stack_cfg, dim=dim, use_adanorm=style_emb_mode == EmbeddingModes.ADANORM, style_emb_dim=self.style_emb_dim)
        self.pos_emb = AbsolutePositionalEmbedding(dim, max_seq_len) if use_abs_pos_emb else None
        if self.pos_emb is not None:
            nn.init.kaiming_normal_(self.pos_emb.emb.weight)
        self.emb_norm = LayerNorm(dim) if emb_norm else nn.Identity()
        self.emb_dropout = nn.Dropout(emb_dropout) if emb_dropout > 0. else nn.Identity()   # (the forward runs F_.dropout with its p)
        width_in = dim + (self.context_emb_dim if cat_context else 0) + (self.style_emb_dim if cat_style else 0)
        self.project_emb = nn.Linear(width_in, dim) if width_in != dim else nn.Identity()
        self.lm_head = (TupleTokenHeadsRegistry.instantiate(config=lm_head, dim=dim, embeddings=self.token_emb)
                        if lm_head is not None else None)
        self.regression_head = None
        if regression_head is not None:
            assert self.token_emb.continuous, "TupleTokenRegressionHead depends on `continuous` token embeddings."
            self.regression_head = TupleTokenRegressionHead.init(config=regression_head, dim=dim)

    @no_autocast
    def forward(self, x: Tensor, mask: Optional[Tensor] = None, x_extra=None, style_embeddings: Optional[Tensor] = None,
                context: Optional[Tensor] = None, context_mask: Optional[Tensor] = None,
                caches: Optional[TupleTransformerCaches] = None, logits_keys: Optional[List] = None,
                return_embeddings: bool = False, return_attn: bool = False, return_caches: bool = False,
                labels: Optional[Tensor] = None, want_argmax: bool = False, label_counts=None, **kwargs):
        token_emb_cache = caches.token_emb if caches is not None else None
        if hasattr(self.token_emb, "multiseq_mode") and x_extra is not None:
            x_extra = [x_extra] if isinstance(x_extra, Tensor) else x_extra
            token_emb = self.token_emb([x] + list(x_extra), cache=token_emb_cache)
        else:
            token_emb = self.token_emb(x, cache=token_emb_cache)
        x = token_emb
        if self.pos_emb is not None:
            x = x + self.pos_emb(x).to(x.dtype)
        if isinstance(self.emb_norm, nn.LayerNorm):
            # encoders (nothing concatenated, no input projection): the norm writes the fp32 residual stream itself -- as bf16 it was cast
            # up by the layer stack right away (one pass forward, one backward, per encoder)
            will_cat = ((context is not None and self.context_emb_mode == EmbeddingModes.CONCAT)
                        or (style_embeddings is not None and self.style_emb_mode == EmbeddingModes.CONCAT))
            x = self.emb_norm(x, out_fp32=not will_cat and not isinstance(self.project_emb, nn.Linear))
        parts = [x]
        if context is not None and self.context_emb_mode == EmbeddingModes.CONCAT:
            parts.append(context[:, :x.shape[1]])
            context = None
        if style_embeddings is not None:
            style_embeddings = style_embeddings[:, :x.shape[1]]
            if self.style_emb_mode == EmbeddingModes.CONCAT:
                parts.append(style_embeddings)
                style_embeddings = None
        if len(parts) > 1:
            x = F_.cat_cast(*parts)
        x = F_.dropout(x, getattr(self.emb_dropout, "p", 0.0), self.training)      # behind the concatenation (transformer.py:184; 0 in every shipped recipe)
        if isinstance(self.project_emb, nn.Linear):
            x = F_.linear(x, self.project_emb.weight, self.project_emb.bias, out_fp32=True)

        out, intermediates = self.transformer(
            x, mask=mask.contiguous() if mask is not None else None, context=context, context_mask=context_mask,
            style_embeddings=style_embeddings, intermediates_cache=caches.transformer if caches is not None else None,
            return_hiddens=True)

        logits = ce_sums = argmax = eval_sums = None
        if not return_embeddings and self.lm_head is not None:
            res = self.lm_head(out, keys=logits_keys, labels=labels, want_argmax=want_argmax, label_counts=label_counts)
            if isinstance(res, tuple):
                logits, ce_sums, argmax = res
                eval_sums = getattr(self.lm_head, "ce_state", {}).get("metrics")
            else:
                logits = res
        reg_values = None
        if not return_embeddings and self.regression_head is not None:
            reg_values = self.regression_head(out, keys=logits_keys)
        if return_attn:  # quirk (ii) of the reference: it reads a field that does not exist (transformer.py:205-207)
            raise AttributeError("'AttentionIntermediates' object has no attribute 'post_softmax_attn'")
        out_caches = TupleTransformerCaches(token_emb=token_emb, transformer=intermediates) if return_caches else None
        return TupleTransformerOutput(hidden_state=out, logits=logits, attentions=None, caches=out_caches,
                                      reg_values=reg_values, ce_sums=ce_sums, argmax=argmax, eval_sums=eval_sums)
