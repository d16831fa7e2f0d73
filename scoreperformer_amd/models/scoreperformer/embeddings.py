"""Tuple-token embeddings and LM heads with the reference's contract (`models/scoreperformer/embeddings.py:31-462`).

HIP execution: all per-key tables are built in one launch and shared (per forward pass) by the three transformers and the
tied LM head; gather + concat + LayerNorm is one kernel; projections are MFMA GEMMs; the tied LM head is one GEMM to the
concatenated embedding space, one LayerNorm and a per-key GEMM + fused cross-entropy.
"""
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Union, Dict, Optional, List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F_
from ...modules.constructor import Constructor, Registry, VariableModuleConfig
from ...modules.transformer.embeddings import DiscreteContinuousEmbedding, DiscreteDenseContinuousEmbedding
from ...modules.layers import LayerNorm
from ...utils.config import MISSING
from ...utils.amp import no_autocast

# ---------------------------------------------------------------------------------------------------------
# table building (K1), shared per forward pass
# ---------------------------------------------------------------------------------------------------------

_TABLE_CACHE = {"enabled": False, "tables": {}}


@contextmanager
def shared_tables():
    """Within this context every per-key table is built once and reused (the reference rebuilds `emb.weight` at each of
    its 5 uses per step, modules/transformer/embeddings.py:158-163)."""
    outer = _TABLE_CACHE["enabled"]
    if not outer:
        _TABLE_CACHE["enabled"], _TABLE_CACHE["tables"] = True, {}
    try:
        yield
    finally:
        if not outer:
            _TABLE_CACHE["enabled"], _TABLE_CACHE["tables"] = False, {}


def build_tables(embs: List[nn.Module]) -> List[Tensor]:
    """fp32 [V, E] table per embedding module; continuous tables of equal kind are built by one kernel launch."""
    cache = _TABLE_CACHE["tables"] if _TABLE_CACHE["enabled"] else {}
    key = lambda m: (id(m), torch.is_grad_enabled())
    todo = [m for m in dict.fromkeys(embs) if key(m) not in cache]
    groups: Dict[tuple, List[nn.Module]] = {}
    for m in todo:
        if isinstance(m, DiscreteContinuousEmbedding) and m.continuous:
            groups.setdefault((m.dense, m.discrete, m.has_discrete, m.ids_mask), []).append(m)
        else:
            cache[key(m)] = m.index_weight if isinstance(m, DiscreteContinuousEmbedding) else m.weight
    for (dense, discrete, has_iw, ids_mask), mods in groups.items():
        for start in range(0, len(mods), 16):
            chunk = mods[start:start + 16]
            cols = list(zip(*[m.table_params() for m in chunk]))  # tv, w0, b0, w1, b1, iw
            flat = list(cols[0]) + list(cols[1])
            if dense:
                flat += list(cols[2]) + list(cols[3]) + list(cols[4])
            if has_iw:
                flat += list(cols[5])
            tables = F_.TableBuildFn.apply(len(chunk), dense, discrete, has_iw, ids_mask, *flat)
            if torch.is_grad_enabled():
                F_.share_table_grads(tables)
            for m, t in zip(chunk, tables):
                cache[key(m)] = t
    return [cache[key(m)] for m in embs]


TupleTokenEmbeddingsRegistry = type("_TupleTokenEmbeddingsRegistry", (Registry,), {})()


@dataclass
class TupleTokenEmbeddingsConfig(VariableModuleConfig):
    _target_: str = "simple"
    num_tokens: Dict[str, int] = MISSING
    emb_dims: Union[Dict[str, int], int] = MISSING
    mode: str = "cat"
    project_emb_dim: int = 512
    emb_norm: bool = False
    discrete: bool = True
    continuous: Union[bool, List[str]] = False
    continuous_dense: bool = False
    token_values: Optional[Dict[str, list]] = None
    discrete_ids: Optional[List[int]] = None
    tie_keys: Optional[Dict[str, str]] = None


@TupleTokenEmbeddingsRegistry.register("simple")
class TupleTokenEmbeddings(nn.Module, Constructor):
    def __init__(self, num_tokens: Dict[str, int], emb_dims: Union[Dict[str, int], int], mode: str = "cat",
                 project_emb_dim: int = 512, emb_norm: bool = False, discrete: bool = True,
                 continuous: Union[bool, List[str]] = False, continuous_dense: bool = False,
                 token_values: Optional[Dict[str, list]] = None, discrete_ids: Optional[List[int]] = None,
                 tie_keys: Optional[Dict[str, str]] = None):
        super().__init__()
        self.mode = mode
        if mode not in ("cat", "sum"):
            raise ValueError(f"TupleTokenEmbeddings mode {mode!r}: 'cat' or 'sum'")
        if mode == "sum":
            assert isinstance(emb_dims, int) or all(e == list(emb_dims.values())[0] for e in emb_dims.values()), \
                "`emb_dims` in TupleTokenEmbeddings' `sum` mode should be the same for all keys."
        continuous_keys = continuous
        if isinstance(continuous, bool):
            continuous_keys = [key for key in num_tokens] if continuous else []
        else:
            continuous_keys = list(continuous)
            continuous = len(continuous_keys) > 0
        total_emb_dim = 0
        embeddings = {}
        token_values = token_values or {}
        for key, num in num_tokens.items():
            emb_dim = emb_dims if isinstance(emb_dims, int) else emb_dims[key]
            if tie_keys and key in tie_keys:
                embeddings[key] = embeddings[tie_keys[key]]
                emb_dim = emb_dims if isinstance(emb_dims, int) else emb_dims[tie_keys[key]]
            elif key in continuous_keys:
                cls = DiscreteDenseContinuousEmbedding if continuous_dense else DiscreteContinuousEmbedding
                embeddings[key] = cls(num_embeddings=num, embedding_dim=emb_dim, discrete=discrete, continuous=True,
                                      discrete_ids=list(discrete_ids) if discrete_ids is not None else None,
                                      token_values=token_values.get(key, None), padding_idx=0)
            else:
                embeddings[key] = nn.Embedding(num, emb_dim, padding_idx=0)
            total_emb_dim += emb_dim if mode == "cat" else emb_dim - total_emb_dim     # `sum`: the common width (embeddings.py:117)
        self.embs = nn.ModuleDict(embeddings)
        self.norm = LayerNorm(total_emb_dim) if emb_norm else nn.Identity()
        if total_emb_dim != project_emb_dim:
            self.project_emb = nn.Linear(total_emb_dim, project_emb_dim)
        self.num_tokens = dict(num_tokens)
        self.emb_dims = emb_dims
        self.total_emb_dim = total_emb_dim
        self.continuous = continuous
        self.continuous_keys = continuous_keys
        self.token_values = token_values
        self.init_()

    def init_(self):
        if not self.continuous:
            for key, emb in self.embs.items():
                weight_attr = "index_weight" if key in self.continuous_keys else "weight"
                nn.init.kaiming_normal_(getattr(emb, weight_attr))

    def tables(self) -> List[Tensor]:
        return build_tables(list(self.embs.values()))

    def _forward_project(self, tokens: Union[Tensor, List[Tensor]]) -> Tensor:
        """gather + concat + LayerNorm (one kernel) and the projection GEMM; tokens int64 [b, n, >= K] (any strides).
        A LIST of token tensors is the multi-sequence `pre-sum` form (embeddings.py:231-241): the per-key embeddings of the sequences are
        summed before norm / projection.  `sum` mode (embeddings.py:141): the per-key embeddings are summed instead of concatenated and
        only normalised.  Both run as gather kernels without the norm (one launch per sequence), an fp32 sum and the LayerNorm kernel --
        no shipped recipe takes them, so they are not fused further."""
        has_norm = isinstance(self.norm, nn.LayerNorm)
        seqs = list(tokens) if isinstance(tokens, (list, tuple)) else [tokens]
        if self.mode == "cat" and len(seqs) == 1:
            e = F_.EmbedFn.apply(seqs[0], self.norm.weight if has_norm else None, self.norm.bias if has_norm else None,
                                 self.norm.eps if has_norm else 1e-5, *self.tables())
            # quirk kept from the reference: `cat` mode calls project_emb unconditionally (embeddings.py:139)
            return F_.linear(e, self.project_emb.weight, self.project_emb.bias)
        tabs = self.tables()
        e = None
        for t in seqs:                                  # [b, n, sum E] per sequence, no norm
            part = F_.EmbedFn.apply(t, None, None, 1e-5, *tabs).float()
            e = part if e is None else e + part
        if self.mode == "sum":                          # sum over the keys: [b, n, K, E] -> [b, n, E]
            e = e.unflatten(-1, (len(tabs), -1)).sum(dim=-2)
        if has_norm:
            e = F_.layer_norm(e, self.norm.weight, self.norm.bias, eps=self.norm.eps, out_fp32=self.mode == "sum")
        if self.mode == "sum":
            return e
        return F_.linear(e, self.project_emb.weight, self.project_emb.bias)

    @no_autocast
    def forward(self, x: Tensor, values: Optional[Tensor] = None, cache: Optional[Tensor] = None,
                return_embeddings: bool = False):
        if values is not None or return_embeddings:
            raise NotImplementedError("`values` / `return_embeddings` are not used on the ScorePerformer hot path")
        if cache is not None:
            x = x[:, cache.shape[1]:]
        token_emb = self._forward_project(x)
        if cache is not None:
            token_emb = torch.cat([cache, token_emb], dim=1)
        return token_emb


@dataclass
class MultiSeqTupleTokenEmbeddingsConfig(TupleTokenEmbeddingsConfig):
    _target_: str = "multi-seq"
    multiseq_mode: str = "pre-sum"
    num_sequences: int = 2


@TupleTokenEmbeddingsRegistry.register("multi-seq")
class MultiSeqTupleTokenEmbeddings(TupleTokenEmbeddings):
    def __init__(self, num_tokens, emb_dims, mode: str = "cat", project_emb_dim: int = 512, emb_norm: bool = False,
                 discrete: bool = True, continuous=False, continuous_dense: bool = False, token_values=None,
                 discrete_ids=None, tie_keys=None, multiseq_mode: str = "pre-sum", num_sequences: int = 2):
        super().__init__(num_tokens=num_tokens, emb_dims=emb_dims, mode=mode, project_emb_dim=project_emb_dim,
                         emb_norm=emb_norm, discrete=discrete, continuous=continuous, continuous_dense=continuous_dense,
                         token_values=token_values, discrete_ids=discrete_ids, tie_keys=tie_keys)
        self.multiseq_mode = multiseq_mode
        self.num_sequences = num_sequences
        if self.multiseq_mode == "post-cat":
            self.project_multiemb = nn.Linear(num_sequences * project_emb_dim, project_emb_dim)

    @no_autocast
    def forward(self, tokens: Union[Tensor, List[Tensor]], values=None, cache: Optional[Tensor] = None,
                return_embeddings: bool = False):
        if isinstance(tokens, (list, tuple)) and len(tokens) == 1:
            tokens = tokens[0]
        if isinstance(tokens, Tensor):
            return super().forward(tokens, values=values, cache=cache, return_embeddings=return_embeddings)
        if values is not None or return_embeddings:
            raise NotImplementedError("`values` / `return_embeddings` are not used on the ScorePerformer hot path")
        if cache is not None:
            tokens = [t[:, cache.shape[1]:] for t in tokens]
        if self.multiseq_mode == "post-cat":
            assert len(tokens) == self.num_sequences
            proj = [self._forward_project(t) for t in tokens]
            token_emb = F_.linear(F_.cat_cast(*proj), self.project_multiemb.weight, self.project_multiemb.bias)
        elif self.multiseq_mode.startswith("post"):
            proj = [self._forward_project(t) for t in tokens]
            token_emb = proj[0]
            for p in proj[1:]:
                token_emb = token_emb + p
        elif self.multiseq_mode == "pre-sum":
            token_emb = self._forward_project(list(tokens))
        else:
            # the reference returns None here (embeddings.py:257-258) and its caller then dies on `None + Tensor`; same outcome, said clearly.
            # Raised where several sequences actually meet the unknown mode, not at construction: a single sequence never looks at it.
            raise ValueError(f"MultiSeqTupleTokenEmbeddings: unknown multiseq_mode {self.multiseq_mode!r} "
                             "(expected 'pre-sum', 'post-sum' / any 'post*', or 'post-cat')")
        if cache is not None:
            token_emb = torch.cat([cache, token_emb], dim=1)
        return token_emb


# ---------------------------------------------------------------------------------------------------------
# heads
# ---------------------------------------------------------------------------------------------------------

TupleTokenHeadsRegistry = type("_TupleTokenHeadsRegistry", (Registry,), {})()


@dataclass
class TupleTokenHeadsConfig(VariableModuleConfig):
    dim: int = MISSING


@dataclass
class TupleTokenLMHeadConfig(TupleTokenHeadsConfig):
    _target_: str = "lm"
    num_tokens: Optional[Dict[str, int]] = None
    embeddings: Optional[TupleTokenEmbeddings] = None
    filter_keys: Optional[List[str]] = None


def _wanted(i, key, keys):
    return keys is None or i in keys or key in keys


class LazyLogits(dict):
    """{key: logits} in the reference's key order.  Keys whose label column holds no valid label contribute nothing to the
    training loss, so their logits GEMM is deferred until somebody actually reads the entry."""
    _PENDING = object()

    def defer(self, key, thunk):
        dict.__setitem__(self, key, LazyLogits._PENDING)
        self.__dict__.setdefault("_thunks", {})[key] = thunk

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is LazyLogits._PENDING:
            v = self._thunks.pop(key)()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _force(self):
        for key in list(getattr(self, "_thunks", {})):
            self[key]

    def items(self):
        self._force()
        return dict.items(self)

    def values(self):
        self._force()
        return dict.values(self)

    def copy(self):
        self._force()
        return dict(self)


class _HeadBase(nn.Module):
    """`forward(x, keys=None)` -> {key: logits}.  With `labels` (int64 [b, n, K] view, -100 = ignore) the per-key
    cross-entropy is fused behind the logits GEMM: returns (logits, sums) with sums[key] = (loss_sum, valid_count).
    `label_counts` = (pinned int32 [K] valid-label counts, event) issued by the LM wrapper before the decoder ran: keys
    with a zero count skip their GEMM + cross-entropy in forward AND backward (the reference drops them from the loss,
    wrappers.py:56)."""

    def _per_key(self, items, labels, ignore_index, want_argmax, label_counts=None):
        logits, sums, argmax = LazyLogits(), {}, {}
        self.ce_state = {"active": None, "eval": getattr(self, "eval_spec", None)}   # eval_spec: set by ScorePerformerEvaluator.attach()
        counts = None
        if label_counts is not None and labels is not None:
            buf, ev = label_counts
            ev.synchronize()   # recorded before the decoder layers were enqueued: long complete, no pipeline bubble
            counts = buf.tolist()
            self.ce_state["active"] = {key for i, key, *_ in items if counts[i] > 0}
            self.ce_state["counts"] = counts
        for i, key, e, table, bias in items:
            if counts is not None and counts[i] == 0 and not want_argmax:
                def thunk(e=e, table=table, bias=bias):
                    lg, _, _ = F_.HeadCEFn.apply(e, table, bias, None, ignore_index, False, None, None)
                    return lg.view(*e.shape[:-1], lg.shape[-1])
                logits.defer(key, thunk)
                sums[key] = F_.ops.zeros_small(2, e.device)
                continue
            lab = labels[..., i] if labels is not None else None
            lg, sm, am = F_.HeadCEFn.apply(e, table, bias, lab, ignore_index, want_argmax, self.ce_state, key)
            logits[key] = lg.view(*e.shape[:-1], lg.shape[-1])
            if sm is not None:
                sums[key] = sm
            if am is not None:
                argmax[key] = am.view(e.shape[:-1])
        return logits, sums, argmax


@TupleTokenHeadsRegistry.register("lm")
class TupleTokenLMHead(_HeadBase, Constructor):
    def __init__(self, dim: int, num_tokens: Optional[Dict[str, int]] = None, embeddings: Optional[TupleTokenEmbeddings] = None,
                 filter_keys: Optional[List[str]] = None):
        assert num_tokens is not None or embeddings is not None
        super().__init__()
        num_tokens = num_tokens or embeddings.num_tokens
        self.heads = nn.ModuleDict({key: nn.Linear(dim, num) for key, num in num_tokens.items()
                                    if not filter_keys or key in filter_keys})

    @no_autocast
    def forward(self, x: Tensor, keys=None, labels: Optional[Tensor] = None, ignore_index: int = -100, want_argmax=False,
                label_counts=None):
        xb = F_.cast(x, torch.bfloat16)
        items = [(i, key, xb, head.weight, head.bias) for i, (key, head) in enumerate(self.heads.items())
                 if _wanted(i, key, keys)]
        logits, sums, argmax = self._per_key(items, labels, ignore_index, want_argmax, label_counts)
        return (logits, sums, argmax) if labels is not None or want_argmax else logits


@dataclass
class TupleTokenTiedLMHeadConfig(TupleTokenHeadsConfig):
    _target_: str = "lm-tied"
    embeddings: TupleTokenEmbeddings = MISSING
    reuse_projection: bool = True


@TupleTokenHeadsRegistry.register("lm-tied")
class TupleTokenTiedLMHead(_HeadBase, Constructor):
    def __init__(self, dim: int, embeddings: TupleTokenEmbeddings, reuse_projection: bool = True):
        super().__init__()
        self.embs = embeddings.embs
        self.total_emb_dim = embeddings.total_emb_dim
        self.split_dims = [emb.embedding_dim for emb in embeddings.embs.values()]
        if reuse_projection:
            assert dim == embeddings.project_emb.out_features, \
                f"Projection layer could be reused only if last input tensor dimension " \
                f"is equal to projection layer's `out_features = {embeddings.project_emb.out_features}`"
            self.project_emb = embeddings.project_emb
        else:
            self.project_emb = nn.Linear(dim, self.total_emb_dim, bias=False)
        self.norm = LayerNorm(self.total_emb_dim)
        self.reuse_projection = reuse_projection

    @no_autocast
    def forward(self, x: Tensor, keys=None, labels: Optional[Tensor] = None, ignore_index: int = -100, want_argmax=False,
                label_counts=None):
        # `x @ project_emb.weight` uses the weight UNtransposed (embeddings.py:346) -- the embeddings' [dim, total] projection when
        # it is reused, and equally the head's own Linear(dim, total) weight [total, dim] otherwise: like the reference, that second
        # form only multiplies when dim == total, and a checkpoint of it gives the same logits here
        e = self.norm(F_.linear(x, self.project_emb.weight, kn_layout=True))
        tables = build_tables(list(self.embs.values()))
        parts = F_.split_cols(e, self.split_dims) if e.requires_grad else torch.split(e, self.split_dims, dim=-1)
        items = [(i, key, parts[i], tables[i], None) for i, key in enumerate(self.embs.keys()) if _wanted(i, key, keys)]
        logits, sums, argmax = self._per_key(items, labels, ignore_index, want_argmax, label_counts)
        return (logits, sums, argmax) if labels is not None or want_argmax else logits


@dataclass
class TupleTokenTiedSplitLMHeadConfig(TupleTokenHeadsConfig):
    _target_: str = "lm-tied-split"
    embeddings: TupleTokenEmbeddings = MISSING
    filter_keys: Optional[List[str]] = None


@TupleTokenHeadsRegistry.register("lm-tied-split")
class TupleTokenTiedSplitLMHead(_HeadBase, Constructor):
    def __init__(self, dim: int, embeddings: TupleTokenEmbeddings, filter_keys: Optional[List[str]] = None):
        super().__init__()
        to_embs = {}
        for key, token_emb in embeddings.embs.items():
            if not filter_keys or key in filter_keys:
                to_embs[key] = nn.Sequential(nn.Linear(dim, token_emb.embedding_dim), LayerNorm(token_emb.embedding_dim))
        self.to_embs = nn.ModuleDict(to_embs)
        self.embs = embeddings.embs

    @no_autocast
    def forward(self, x: Tensor, keys=None, labels: Optional[Tensor] = None, ignore_index: int = -100, want_argmax=False,
                label_counts=None):
        tables = build_tables(list(self.embs.values()))
        items = []
        for i, key in enumerate(self.embs.keys()):
            if _wanted(i, key, keys):
                lin, ln = self.to_embs[key]
                items.append((i, key, ln(F_.linear(x, lin.weight, lin.bias)), tables[i], None))
        logits, sums, argmax = self._per_key(items, labels, ignore_index, want_argmax, label_counts)
        return (logits, sums, argmax) if labels is not None or want_argmax else logits


@dataclass
class TupleTokenRegressionHeadConfig(TupleTokenHeadsConfig):
    _target_: str = "regression"
    regression_keys: List[str] = MISSING


@TupleTokenHeadsRegistry.register("regression")
class TupleTokenRegressionHead(nn.Module, Constructor):
    def __init__(self, dim: int, regression_keys: List[str]):
        super().__init__()
        self.layers = nn.ModuleDict({key: nn.Linear(dim, 1) for key in regression_keys})

    @no_autocast
    def forward(self, x: Tensor, keys=None):
        return {key: F_.linear_f32(F_.cast(x, torch.float32), layer.weight, layer.bias)
                for i, (key, layer) in enumerate(self.layers.items()) if _wanted(i, key, keys)}


@dataclass
class TupleTokenEmbeddingHeadConfig(TupleTokenHeadsConfig):
    _target_: str = "embedding"
    emb_dim: int = MISSING
    hidden_dim: Optional[int] = None
    depth: int = 2
    detach_inputs: Union[bool, float] = True


@TupleTokenHeadsRegistry.register("embedding")
class TupleTokenEmbeddingHead(nn.Module, Constructor):
    """Hidden state -> an embedding vector through a Mish MLP (`models/scoreperformer/embeddings.py:424-462`); `detach_inputs` blends
    the detached and the attached input (1 / True: no gradient reaches the transformer).  state_dict keys `layers.<2i>.{weight,bias}`
    as in the reference's nn.Sequential of Linear / Mish pairs.  Small fp32 contractions: exact-fp32 GEMM + the Mish kernel."""

    def __init__(self, dim: int, emb_dim: int, hidden_dim: Optional[int] = None, depth: int = 2,
                 detach_inputs: Union[bool, float] = True):
        super().__init__()
        hidden_dim = hidden_dim or emb_dim
        widths = [dim] + [hidden_dim] * (depth - 1) + [emb_dim]
        stack = []
        for level in range(depth):
            stack.append(nn.Linear(widths[level], widths[level + 1]))
            if level + 1 < depth:
                stack.append(nn.Mish())
        self.layers = nn.Sequential(*stack)
        self.detach_inputs = detach_inputs

    @no_autocast
    def forward(self, x: Tensor):
        keep = float(self.detach_inputs)
        x = F_.cast(x, torch.float32)
        if keep >= 1.0:
            h = x.detach()
        elif keep <= 0.0:
            h = x
        else:
            h = keep * x.detach() + (1.0 - keep) * x
        for layer in self.layers:
            h = F_.linear_f32(h, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else F_.mish(h)
        return h

