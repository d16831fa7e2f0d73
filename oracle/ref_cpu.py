"""TEST INFRASTRUCTURE ONLY — functional fp32 CPU restatement of the ScorePerformer hot path.

Every function takes the *reference's* flat ``state_dict`` (same keys as
``ScorePerformer.state_dict()``) plus the dict config and plain
tensors, and re-derives the reference result with explicit math (no
``nn.Module``s, no SDPA).  Line citations are file-local line numbers under
``/root/reference/scoreperformer/``.

Parity status: PINNED against the reference itself — see
``tests/test_oracle_golden.py`` and ``oracle/refimport/make_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

SEGMENT_MODES = ("bar_mean", "beat_mean", "onset_mean", "isolated_bar_mean")


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    try:
        v = cfg.get(key, default)
    except AttributeError:
        v = getattr(cfg, key, default)
    return default if v is None else v


# --------------------------------------------------------------------------------------
# A9  ALiBi slopes / bias                       modules/transformer/embeddings.py:268-325
# --------------------------------------------------------------------------------------

def alibi_slopes(heads: int) -> List[float]:
    """`ALiBiPositionalBias._compute_slopes` (embeddings.py:280-292)."""

    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]

    if math.log2(heads).is_integer():
        return pow2(heads)
    c = 2 ** math.floor(math.log2(heads))
    return pow2(c) + pow2(2 * c)[0::2][:heads - c]


def alibi_bias(i: int, j: int) -> Tensor:
    """`get_bias(i, j, k=j-i)` (embeddings.py:294-297): -(|jj - (ii + j - i)|), shape (1, i, j)."""
    ii = torch.arange(j - i, j)
    jj = torch.arange(j)
    return -(jj[None, None, :] - ii[None, :, None]).abs().float()


# --------------------------------------------------------------------------------------
# A5  LayerNorm / AdaptiveLayerNorm               modules/layers.py:31-47
# --------------------------------------------------------------------------------------

def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def ada_layer_norm(x: Tensor, cond: Tensor, lin_w: Tensor, lin_b: Tensor) -> Tensor:
    """`AdaptiveLayerNorm.forward` (layers.py:41-47): gamma, beta = Linear(cond).chunk(2)."""
    gb = cond @ lin_w.t() + lin_b
    gamma, beta = gb.chunk(2, dim=-1)
    return gamma * layer_norm(x, None, None) + beta


def norm(sd: SD, prefix: str, x: Tensor, cond: Optional[Tensor], ada: bool) -> Tensor:
    if ada:
        return ada_layer_norm(x, cond, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    return layer_norm(x, sd[prefix + "weight"], sd[prefix + "bias"])


# --------------------------------------------------------------------------------------
# A6/A7  Attention                 modules/transformer/attention.py:107-222, attend.py:58-126
# --------------------------------------------------------------------------------------

# Dropout in a training-mode comparison.  torch's dropout draws cannot be reproduced by a kernel that derives its masks from counters, so
# the comparison goes the other way: the test reads the masks the product used (attention probabilities: [b, h, i, j]; feed-forward
# activations: [.., inner]) and this oracle applies exactly those -- DROP_FEED(kind, prefix, tensor) returns tensor * keep / (1 - p) for
# the module whose state_dict prefix is `prefix` (kind "attn": attend.py:122; "ffn": feedforward.py:57-60).  None = no dropout (p = 0).
DROP_FEED = None


def attention(
        sd: SD, prefix: str, x: Tensor, *, heads: int, causal: bool, alibi: bool = True,
        context: Optional[Tensor] = None, mask: Optional[Tensor] = None, context_mask: Optional[Tensor] = None,
        cache_k: Optional[Tensor] = None, cache_v: Optional[Tensor] = None, return_kv: bool = False,
        single_query_bias: bool = False):
    """q/k/v projections (attention.py:135-142), key-padding + causal + ALiBi folded into one additive
    mask (attend.py:80-113), softmax(q k^T * dh^-0.5 + bias) v, out-proj and query-row masking
    (attention.py:210-218).  MQA when `to_k.weight` has dim_head rows (attention.py:67-73)."""
    b, n, _ = x.shape
    kv_in = x if context is None else context
    wq, wk, wv, wo = (sd[prefix + f"to_{t}.weight"] for t in ("q", "k", "v", "out"))
    dh = wq.shape[0] // heads
    mqa = wk.shape[0] == dh
    q = (x @ wq.t()).view(b, n, heads, dh).transpose(1, 2)  # b h n d
    k = kv_in @ wk.t()
    v = kv_in @ wv.t()
    if cache_k is not None:  # attention.py:155-156
        k = torch.cat([cache_k, k], dim=-2)
        v = torch.cat([cache_v, v], dim=-2)
    kv_cache = (k, v)
    if not mqa:
        if k.ndim == 3:
            k = k.view(b, -1, heads, dh).transpose(1, 2)
            v = v.view(b, -1, heads, dh).transpose(1, 2)
    else:
        k = k.unsqueeze(1)
        v = v.unsqueeze(1)
    key_mask = mask if context_mask is None else context_mask  # attention.py:144
    if prefix + "mem_k" in sd:   # attention.py:146-153: learned memories [h, m, d] in front of the keys / values, always visible
        assert not mqa and cache_k is None
        mem_k, mem_v = sd[prefix + "mem_k"], sd[prefix + "mem_v"]
        k = torch.cat([mem_k[None].expand(b, -1, -1, -1), k], dim=-2)
        v = torch.cat([mem_v[None].expand(b, -1, -1, -1), v], dim=-2)
        if key_mask is not None:
            key_mask = F.pad(key_mask, (mem_k.shape[1], 0), value=True)
    j = k.shape[-2]

    dots = (q @ k.transpose(-1, -2)) * dh ** -0.5  # b h i j
    allowed = torch.ones(b, 1, n, j, dtype=torch.bool)
    if key_mask is not None:
        allowed = allowed & key_mask[:, None, None, :]
    if causal:
        allowed = allowed & ~torch.ones(n, j, dtype=torch.bool).triu(j - n + 1)

    slopes_key = prefix + "rel_pos.learned_logslopes"
    if slopes_key in sd:
        slopes = sd[slopes_key].exp().view(-1, 1, 1)  # embeddings.py:324-325
        has_alibi = True
    else:
        has_alibi = alibi
        slopes = torch.tensor(alibi_slopes(heads)).view(-1, 1, 1) if alibi else None
    if has_alibi:
        if slopes.shape[0] < heads:  # embeddings.py:307-308
            slopes = F.pad(slopes, (0, 0, 0, 0, 0, heads - slopes.shape[0]))
        # single_query_bias: the cached decode calls a cross-attention block with ONE query (transformer.py:159,201), so every position's
        # bias row was computed as get_bias(i = 1, j, k = j - 1) = -|key - (j - 1)| when that position was the newest one
        dots = dots + slopes[None] * (alibi_bias(1, j) if single_query_bias else alibi_bias(n, j))[None]
    neg = -torch.finfo(dots.dtype).max // 2  # attend.py:102,105
    dots = torch.where(allowed, dots, torch.full_like(dots, neg))
    attn = dots.softmax(dim=-1)
    if DROP_FEED is not None:  # attend.py:122: F.dropout on the probabilities, with the masks of the run under test (see DROP_FEED)
        attn = DROP_FEED("attn", prefix, attn)
    out = attn @ v  # b h i d
    out = out.transpose(1, 2).reshape(b, n, heads * dh) @ wo.t()
    if mask is not None:  # attention.py:216-218
        qmask = mask[:, -1:] if cache_k is not None else mask
        out = out * qmask[..., None]
    if return_kv:
        return out, kv_cache
    return out


# --------------------------------------------------------------------------------------
# A8  FeedForward                               modules/transformer/feedforward.py:13-64
# --------------------------------------------------------------------------------------

def feed_forward(sd: SD, prefix: str, x: Tensor, *, glu: bool, swish: bool) -> Tensor:
    act = F.silu if swish else F.gelu
    if glu:
        h = x @ sd[prefix + "ff.0.proj.weight"].t() + sd[prefix + "ff.0.proj.bias"]
        a, gate = h.chunk(2, dim=-1)
        h = a * act(gate)
    else:
        h = x @ sd[prefix + "ff.0.0.weight"].t()
        if prefix + "ff.0.0.bias" in sd:
            h = h + sd[prefix + "ff.0.0.bias"]
        h = act(h)
    if prefix + "ff.1.weight" in sd:  # post_act_ln
        h = layer_norm(h, sd[prefix + "ff.1.weight"], sd[prefix + "ff.1.bias"])
    if DROP_FEED is not None:  # feedforward.py:57-60: ff.2 = nn.Dropout between the (normalised) activation and the output projection
        h = DROP_FEED("ffn", prefix, h)
    out = h @ sd[prefix + "ff.3.weight"].t()
    if prefix + "ff.3.bias" in sd:
        out = out + sd[prefix + "ff.3.bias"]
    return out


# --------------------------------------------------------------------------------------
# A4  Transformer layer loop                 modules/transformer/transformer.py:139-232
# --------------------------------------------------------------------------------------

def layer_types(tcfg, cross_attend: bool) -> List[str]:
    depth = int(_get(tcfg, "depth", 4))
    only_cross = bool(_get(tcfg, "only_cross", False))
    block = ("a", "c", "f") if cross_attend and not only_cross else ("c", "f") if cross_attend else ("a", "f")
    return list(block) * depth


def transformer(
        sd: SD, prefix: str, x: Tensor, tcfg, *, causal: bool, cross_attend: bool, ada: bool,
        mask=None, context=None, context_mask=None, style=None, return_hiddens: bool = False, cached_decode: bool = False):
    heads = int(_get(tcfg, "heads", 8))
    ff = _get(tcfg, "feed_forward", {})
    glu, swish = bool(_get(ff, "glu", False)), bool(_get(ff, "swish", False))
    alibi = bool(_get(_get(tcfg, "attention", {}), "alibi_pos_bias", False))
    hiddens, kvs = [], []
    for ind, lt in enumerate(layer_types(tcfg, cross_attend)):
        lp = f"{prefix}layers.{ind}."
        if lt == "a":
            hiddens.append(x)
        residual = x
        h = norm(sd, lp + "0.0.", x, style, ada)
        if lt == "a":
            out, kv = attention(sd, lp + "1.", h, heads=heads, causal=causal, alibi=alibi, mask=mask,
                                return_kv=True)
            kvs.append(kv)
        elif lt == "c":
            out, kv = attention(sd, lp + "1.", h, heads=heads, causal=False, alibi=alibi, context=context,
                                mask=mask, context_mask=context_mask, return_kv=True, single_query_bias=cached_decode)
            kvs.append(kv)
        else:
            out = feed_forward(sd, lp + "1.", h, glu=glu, swish=swish)
        x = out + residual
    x = norm(sd, prefix + "final_norm.", x, style, ada)
    hiddens.append(x)
    if return_hiddens:
        return x, hiddens, kvs
    return x


# --------------------------------------------------------------------------------------
# A1  per-key embedding tables            modules/transformer/embeddings.py:118-152,172-222
# --------------------------------------------------------------------------------------

def table_weight(sd: SD, prefix: str, te_cfg) -> Tensor:
    """`DiscreteContinuousEmbedding.weight`: rows `discrete_ids` <- index_weight, all other rows <-
    value MLP(token_values) (dense: Linear(1,E)+Mish -> Linear(E,E); embeddings.py:202-213)."""
    if prefix + "weight" in sd:  # plain nn.Embedding (ablation/no_cont_tokens.yaml)
        return sd[prefix + "weight"]
    discrete = bool(_get(te_cfg, "discrete", True))
    ids = _get(te_cfg, "discrete_ids", None)
    tv = sd[prefix + "token_values"]
    if prefix + "value_layer.0.0.weight" in sd:
        h = tv @ sd[prefix + "value_layer.0.0.weight"].t() + sd[prefix + "value_layer.0.0.bias"]
        i = 1
        while prefix + f"value_layer.{i}.0.weight" in sd:
            h = F.mish(h)
            h = h @ sd[prefix + f"value_layer.{i}.0.weight"].t() + sd[prefix + f"value_layer.{i}.0.bias"]
            i += 1
    else:
        h = tv @ sd[prefix + "value_layer.weight"].t()
    value_w = h
    if ids is not None:
        keep = torch.ones(h.shape[0], 1)
        keep[list(ids)] = 0.
        value_w = h * keep  # embeddings.py:140-141
    if prefix + "index_weight" not in sd:
        return value_w
    iw = sd[prefix + "index_weight"]
    if discrete:
        token_w = iw
    else:
        sel = torch.zeros(iw.shape[0], 1)
        sel[list(ids)] = 1.
        token_w = iw * sel  # embeddings.py:127-130
    return token_w + value_w


def _emb_keys(sd: SD, prefix: str) -> List[str]:
    seen, keys = set(), []
    plen = len(prefix + "embs.")
    for k in sd:
        if k.startswith(prefix + "embs."):
            name = k[plen:].split(".")[0]
            if name not in seen:
                seen.add(name)
                keys.append(name)
    return keys


# --------------------------------------------------------------------------------------
# A2  TupleTokenEmbeddings            models/scoreperformer/embeddings.py:121-165,213-267
# --------------------------------------------------------------------------------------

def tuple_embed_one(sd: SD, prefix: str, tokens, te_cfg, keys: List[str]) -> Tensor:
    """`tokens`: one [b, n, K] tensor, or a list of them = the multi-sequence `pre-sum` form (embeddings.py:231-241: per-key embeddings of
    the sequences summed before norm / projection)."""
    seqs = list(tokens) if isinstance(tokens, (list, tuple)) else [tokens]
    embs = [sum(F.embedding(t[..., i], table_weight(sd, f"{prefix}embs.{k}.", te_cfg), padding_idx=0) for t in seqs)
            for i, k in enumerate(keys)]
    mode = _get(te_cfg, "mode", "cat")
    if mode == "cat":
        e = torch.cat(embs, dim=-1)
        if prefix + "norm.weight" in sd:
            e = layer_norm(e, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
        return e @ sd[prefix + "project_emb.weight"].t() + sd[prefix + "project_emb.bias"]
    e = sum(embs)
    if prefix + "norm.weight" in sd:
        e = layer_norm(e, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
    return e


def tuple_embed(sd: SD, prefix: str, seqs: List[Tensor], te_cfg, keys: List[str]) -> Tensor:
    if len(seqs) == 1 or _get(te_cfg, "_target_", "simple") != "multi-seq":
        return tuple_embed_one(sd, prefix, seqs[0], te_cfg, keys)
    mm = _get(te_cfg, "multiseq_mode", "pre-sum")
    if mm == "post-cat":
        proj = [tuple_embed_one(sd, prefix, t, te_cfg, keys) for t in seqs]
        return torch.cat(proj, dim=-1) @ sd[prefix + "project_multiemb.weight"].t() \
            + sd[prefix + "project_multiemb.bias"]
    if mm.startswith("post"):
        return sum(tuple_embed_one(sd, prefix, t, te_cfg, keys) for t in seqs)
    if mm == "pre-sum":
        return tuple_embed_one(sd, prefix, list(seqs), te_cfg, keys)
    raise NotImplementedError(mm)


# --------------------------------------------------------------------------------------
# A12  LM heads                         models/scoreperformer/embeddings.py:287-311,322-353
# --------------------------------------------------------------------------------------

def lm_head(sd: SD, prefix: str, x: Tensor, te_cfg, keys: List[str], head_cfg,
            only: Optional[List] = None) -> Dict[str, Tensor]:
    target = _get(head_cfg, "_target_", "lm")
    want = lambda i, k: only is None or i in only or k in only
    if target == "lm":
        return {k: x @ sd[f"{prefix}heads.{k}.weight"].t() + sd[f"{prefix}heads.{k}.bias"]
                for i, k in enumerate(keys) if want(i, k)}
    if target == "lm-tied":
        tables = [table_weight(sd, f"{prefix}embs.{k}.", te_cfg) for k in keys]
        split = [t.shape[1] for t in tables]
        e = layer_norm(x @ sd[prefix + "project_emb.weight"], sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
        parts = e.split(split, dim=-1)
        return {k: parts[i] @ tables[i].t() for i, k in enumerate(keys) if want(i, k)}
    if target == "lm-tied-split":   # per key: LayerNorm(Linear(x)) against that key's table (embeddings.py:364-390)
        out = {}
        for i, k in enumerate(keys):
            if want(i, k) and f"{prefix}to_embs.{k}.0.weight" in sd:
                e = x @ sd[f"{prefix}to_embs.{k}.0.weight"].t() + sd[f"{prefix}to_embs.{k}.0.bias"]
                e = layer_norm(e, sd[f"{prefix}to_embs.{k}.1.weight"], sd[f"{prefix}to_embs.{k}.1.bias"])
                out[k] = e @ table_weight(sd, f"{prefix}embs.{k}.", te_cfg).t()
        return out
    raise NotImplementedError(target)


def regression_values(sd: SD, prefix: str, x: Tensor) -> Dict[str, Tensor]:
    """`TupleTokenRegressionHead.forward` (models/scoreperformer/embeddings.py:400-420): one Linear(dim, 1) per regression key."""
    keys = sorted({k[len(prefix) + len("layers."):].split(".")[0] for k in sd if k.startswith(prefix + "layers.")},
                  key=lambda k: [q for q in sd if q.startswith(prefix + "layers.")].index(f"{prefix}layers.{k}.weight"))
    return {k: x @ sd[f"{prefix}layers.{k}.weight"].t() + sd[f"{prefix}layers.{k}.bias"] for k in keys}


def regression_losses(sd: SD, emb_prefix: str, logits: Dict[str, Tensor], reg_values: Dict[str, Tensor], labels: Tensor):
    """L1 between the regression value and the label's token value over non-special labels (wrappers.py:61-78)."""
    out = {}
    for i, key in enumerate(logits.keys()):
        if key not in reg_values:
            continue
        sel = labels[..., i] > 3
        targets = F.embedding(labels[..., i][sel], sd[f"{emb_prefix}embs.{key}.token_values"])
        out[f"{key}/l1"] = F.l1_loss(reg_values[key][sel], targets)
    return out


# --------------------------------------------------------------------------------------
# A3  TupleTransformer.forward              models/scoreperformer/transformer.py:146-222
# --------------------------------------------------------------------------------------

def tuple_transformer(
        sd: SD, prefix: str, cfg, seqs: List[Tensor], *, causal: bool, mask=None, context=None, context_mask=None,
        style=None, with_logits: bool = False, return_hiddens: bool = False, cached_decode: bool = False):
    te = cfg["token_embeddings"]
    keys = _emb_keys(sd, prefix + "token_emb.")
    x = tuple_embed(sd, prefix + "token_emb.", seqs, te, keys)
    if prefix + "pos_emb.emb.weight" in sd:  # embeddings.py:225-242
        w = sd[prefix + "pos_emb.emb.weight"]
        x = x + w[:x.shape[1]] * w.shape[1] ** -0.5
    if prefix + "emb_norm.weight" in sd:
        x = layer_norm(x, sd[prefix + "emb_norm.weight"], sd[prefix + "emb_norm.bias"])
    ctx_mode = _get(cfg, "context_emb_mode", "attention")
    style_mode = _get(cfg, "style_emb_mode", "cat")
    if context is not None and ctx_mode == "cat":
        x = torch.cat([x, context[:, :x.shape[1]]], dim=-1)
        context = None
    if style is not None:
        style = style[:, :x.shape[1]]
        if style_mode == "cat":
            x = torch.cat([x, style], dim=-1)
            style = None
    if prefix + "project_emb.weight" in sd:
        x = x @ sd[prefix + "project_emb.weight"].t() + sd[prefix + "project_emb.bias"]
    res = transformer(sd, prefix + "transformer.", x, cfg["transformer"], causal=causal,
                      cross_attend=context is not None, ada=style_mode == "adanorm" and style is not None,
                      mask=mask, context=context, context_mask=context_mask, style=style,
                      return_hiddens=return_hiddens, cached_decode=cached_decode)
    out = res[0] if return_hiddens else res
    logits = None
    if with_logits:
        logits = lm_head(sd, prefix + "lm_head.", out, te, keys, _get(cfg, "lm_head", {"_target_": "lm"}))
    if return_hiddens:
        return out, logits, res[1], res[2]
    return out, logits


# --------------------------------------------------------------------------------------
# A11  MMD loss                         models/scoreperformer/mmd_transformer.py:505-534
# --------------------------------------------------------------------------------------

def gaussian_kernel_mean(x: Tensor, y: Tensor) -> Tensor:
    d2 = (x[:, None, :] - y[None, :, :]).pow(2).mean(2) / x.shape[-1]
    return torch.exp(-d2).mean()


def compute_mmd(z: Tensor, y: Tensor) -> Tensor:
    return gaussian_kernel_mean(z, z) + gaussian_kernel_mean(y, y) - 2 * gaussian_kernel_mean(z, y)


# --------------------------------------------------------------------------------------
# A10  hierarchical MMD-VAE heads       models/scoreperformer/mmd_transformer.py:169-368
# --------------------------------------------------------------------------------------

def segment_mean(out: Tensor, segments: Tensor) -> Tensor:
    """Mean of `out` rows per segment id (mmd_transformer.py:330-340), via index_add instead of one-hot matmul."""
    b, t, d = out.shape
    S = int(segments.max()) + 1
    sums = torch.zeros(b, S, d).index_put((torch.arange(b)[:, None].expand(b, t), segments), out, accumulate=True)
    counts = torch.zeros(b, S).index_put((torch.arange(b)[:, None].expand(b, t), segments),
                                         torch.ones(b, t), accumulate=True)
    return sums / counts.clamp(min=1.)[..., None]


def mmd_heads(
        sd: SD, prefix: str, cfg, hidden: Tensor, mask: Tensor, segments: Dict[str, Tensor], deadpan_mask: Tensor,
        z_samples: List[Tensor], training: bool = True, drop_masks: Optional[List[Tensor]] = None):
    """Hierarchical branch of `MMDTupleTransformer.forward` (mmd_transformer.py:210-302) with latent
    dropout masks injected (`drop_masks[i]`: (b, S_i, 1) bool, already *not* inclusive) and z injected."""
    modes = list(cfg["aggregate_mode"])
    hierarchical = bool(_get(cfg, "hierarchical", False))
    loss_w = float(_get(cfg, "loss_weight", 1.0))
    deadpan_zero = bool(_get(cfg, "deadpan_zero_latent", False))
    m3 = mask[..., None]
    out = hidden * m3
    b, t = out.shape[:2]
    bi = torch.arange(b)[:, None].expand(b, t)
    losses, latents, embs, drops = {}, [], [], []
    prior_drop = None
    for i, mode in enumerate(modes):
        w, bias = sd[f"{prefix}vae_head.{mode}.linear.weight"], sd[f"{prefix}vae_head.{mode}.linear.bias"]
        if mode == "mean":
            agg = (out.sum(1) / m3.sum(1)).unsqueeze(1)
            lmask = torch.ones(b, 1, dtype=torch.bool)
        elif mode in SEGMENT_MODES:
            seg = segments[mode]
            agg = segment_mean(out, seg)
            lmask = (agg != 0.).all(-1)  # mmd_transformer.py:342
        else:
            agg, lmask = out, mask
        lat = (agg @ w.t() + bias) * lmask[..., None]
        drop = torch.zeros_like(lmask[..., None])
        if drop_masks is not None and drop_masks[i] is not None and mode != "mean" and training:
            drop = drop_masks[i]
        emb = lat
        if mode == "mean":
            emb = emb.expand(-1, t, -1)
            drop = drop.expand(-1, t, -1)
        elif mode in SEGMENT_MODES:
            emb = emb[bi, seg]
            drop = drop[bi, seg]
        emb = emb * m3
        if training and bool(_get(cfg, "inclusive_latent_dropout", True)):  # mmd_transformer.py:249-253
            prior_drop = drop if prior_drop is None else (prior_drop + drop)
            drop = prior_drop
        latents.append(lat)
        embs.append(emb)
        drops.append(drop.expand_as(emb))
        if hierarchical:      # mmd_transformer.py:255-262
            out = torch.cat([out, emb], dim=-1) if bool(_get(cfg, "hierarchical_with_context", True)) else emb
        y = lat[lmask]
        losses[f"MMD/{mode}"] = loss_w * compute_mmd(z_samples[i], y)
        if deadpan_zero:
            dl = lat[deadpan_mask[:, None] * lmask]
            if torch.any(dl):
                losses[f"MMD/{mode}/deadpan"] = F.mse_loss(dl, torch.zeros_like(dl))
    embeddings = torch.cat(embs, dim=-1) * m3
    full, dm = embeddings, None
    if training:                                                       # mmd_transformer.py:284-291
        dm = torch.cat(drops, dim=-1) * m3 * (~deadpan_mask[:, None, None])
        embeddings = embeddings * (~dm)
    loss = sum(losses.values())
    losses["MMD"] = loss
    return dict(latents=latents, embeddings=embeddings, full_embeddings=full, dropout_mask=dm, loss=loss, losses=losses)


# --------------------------------------------------------------------------------------
# A13  LM loss                                models/scoreperformer/wrappers.py:44-84
# --------------------------------------------------------------------------------------

def lm_losses(logits: Dict[str, Tensor], labels: Tensor, ignore_index: int = -100):
    losses = {}
    for i, (key, lg) in enumerate(logits.items()):
        if torch.any(labels[..., i] != ignore_index):
            losses[key] = F.cross_entropy(lg.reshape(-1, lg.shape[-1]), labels[..., i].reshape(-1),
                                          ignore_index=ignore_index)
    return sum(losses.values()) / len(losses), losses


# --------------------------------------------------------------------------------------
# A14  ScorePerformer.forward (mixlm)          models/scoreperformer/model.py:280-341
# --------------------------------------------------------------------------------------

def score_performer_forward(
        sd: SD, cfg, batch: Dict[str, Tensor], z_samples: List[Tensor], training: bool = True,
        drop_masks: Optional[List[Tensor]] = None):
    dec_prefix = "perf_decoder.model." if any(k.startswith("perf_decoder.model.") for k in sd) else "perf_decoder."
    perf, pmask = batch["perf"], batch["perf_mask"]
    score_emb = None
    if cfg.get("score_encoder") is not None:
        score_emb, _ = tuple_transformer(sd, "score_encoder.", cfg["score_encoder"], [batch["score"]], causal=False,
                                         mask=batch["score_mask"])
    pe_cfg = cfg["perf_encoder"]
    # the style encoder reads the noisy performance when the batch carries one (model.py:296-298)
    enc_in, enc_mask = batch.get("noisy_perf", perf), batch.get("noisy_perf_mask", pmask)
    am = pe_cfg["aggregate_mode"]
    if (am if isinstance(am, str) else list(am)[0]) == "isolated_bar_mean":
        # mmd_transformer.py:186-189: the style encoder reads the bar tokens MASKed (ids above EOS = 3 -> MASK = 1); the block-diagonal
        # attention mask it also builds (190-200) is dropped by TupleTransformer.forward's **kwargs and never reaches the layers
        enc_in = enc_in.clone()
        bar_col = enc_in[..., 0]
        bar_col[bar_col > 3] = 1
    hidden, _ = tuple_transformer(sd, "perf_encoder.", pe_cfg, [enc_in], causal=False, mask=enc_mask)
    segs = {"bar_mean": batch["bars"], "beat_mean": batch["beats"], "onset_mean": batch["onsets"], "isolated_bar_mean": batch["bars"]}
    enc = mmd_heads(sd, "perf_encoder.", pe_cfg, hidden, enc_mask, segs, batch["deadpan_mask"], z_samples,
                    training=training, drop_masks=drop_masks)
    dcfg = cfg["perf_decoder"]
    # MixedLM shift (wrappers.py:409-431)
    seq, seq_masked, labels = perf[:, :-1], batch["masked_perf"][:, 1:], batch["labels"][:, 1:]
    ctx = score_emb
    if ctx is not None and _get(dcfg, "context_emb_mode", "attention") == "cat":
        ctx = ctx[:, 1:]
    style = enc["embeddings"][:, 1:]
    seqs = [seq, seq_masked] if _get(dcfg["token_embeddings"], "_target_", "simple") == "multi-seq" else [seq]
    out, logits = tuple_transformer(sd, dec_prefix, dcfg, seqs, causal=True, mask=pmask[:, :-1], context=ctx,
                                    context_mask=batch.get("score_mask"), style=style, with_logits=True)
    ce, losses = lm_losses(logits, labels)
    if any(k.startswith(dec_prefix + "regression_head.") for k in sd):
        reg = regression_losses(sd, dec_prefix + "token_emb.", logits, regression_values(sd, dec_prefix + "regression_head.", out), labels)
        ce = ce + sum(reg.values()) / len(reg)
        losses = dict(losses, **reg)
    loss = ce + enc["loss"]
    losses = dict(losses, **enc["losses"])
    return dict(loss=loss, losses=losses, logits=logits, hidden_state=out, perf_embeddings=enc["embeddings"],
                score_embeddings=score_emb, latents=enc["latents"], perf_full_embeddings=enc["full_embeddings"],
                perf_dropout_mask=enc["dropout_mask"])


# --------------------------------------------------------------------------------------
# decoder-only `Performer` (models/scoreperformer/model.py:62-122) under the three LM wrappers (wrappers.py:87-99,290-307,409-431)
# --------------------------------------------------------------------------------------

def performer_forward(sd: SD, cfg, inputs: Dict[str, Tensor]):
    """cfg = {transformer: <TupleTransformer config>, mode}; inputs = perf, mask, labels[, masked_perf] as `Performer.forward` takes."""
    mode, tcfg = cfg["mode"], cfg["transformer"]
    prefix = "transformer.model." if any(k.startswith("transformer.model.") for k in sd) else "transformer."
    perf, mask, labels = inputs["perf"], inputs["mask"], inputs["labels"]
    if mode == "mlm":        # no shift: every position predicts its own masked dims
        seqs, causal = [perf], False
    else:                    # next-note prediction: inputs lose the last position, targets the first
        labels, mask = labels[:, 1:], mask[:, :-1]
        seqs = [perf[:, :-1]] + ([inputs["masked_perf"][:, 1:]] if mode == "mixlm" else [])
        causal = True
    causal = _get(tcfg["transformer"], "_target_", "default") == "decoder"
    out, logits = tuple_transformer(sd, prefix, tcfg, seqs, causal=causal, mask=mask, with_logits=True)
    loss, losses = lm_losses(logits, labels)
    return dict(loss=loss, losses=losses, logits=logits, hidden_state=out)


# --------------------------------------------------------------------------------------
# A15  greedy MixedLM unmasking (teacher-free)  models/scoreperformer/wrappers.py:325-407
# --------------------------------------------------------------------------------------

@torch.no_grad()
def greedy_unmask(sd: SD, cfg, tokens: Tensor, tokens_masked: Tensor, context: Tensor, style: Tensor,
                  mask_token_id: int = 1, pad_token_id: int = 0, context_mask: Optional[Tensor] = None,
                  reference_hidden_row_defect: bool = False) -> Tensor:
    """Full-prefix recomputation per step (no caches): for each position idx holding MASK tokens, run the
    shifted decoder on out[:, :idx+1], take logits at idx-1 for the masked dims, ban PAD/MASK ids
    (wrappers.py:368-369) and take the argmax (top_k k=1 + multinomial == argmax, sampling.py:28-59)."""
    # reference_hidden_row_defect (cross-attending decoders only): under the cache protocol the reference calls a 'c' block with ONE
    # query but the whole-prefix `mask` (modules/transformer/transformer.py:201), and `out * mask[..., None]` (attention.py:216-218,
    # has_cache is False there) broadcasts that row to s identical rows; they are appended to the cached final hiddens, whose length
    # after step s is s(s+1)/2 instead of s, and `hidden_state[:, idx - 1]` (wrappers.py:364) then reads the hidden of position t - 1
    # with t the smallest integer with t(t+1)/2 >= s -- a STALE position from s = 3 on.  The flag reproduces exactly that (it pins this
    # oracle to the reference's own tokens, tests/golden/tiny_greedy_xattn.npz); the product implements the evident intent, row idx - 1.
    dec_prefix = "perf_decoder.model." if any(k.startswith("perf_decoder.model.") for k in sd) else "perf_decoder."
    dcfg = cfg["perf_decoder"]
    keys = _emb_keys(sd, dec_prefix + "token_emb.")
    out = tokens.clone()
    first_step = None
    unmask = out == mask_token_id
    ids = torch.where(unmask.any(dim=2))[1]
    for idx in ids.tolist():
        dims = torch.where(unmask[0, idx])[0].tolist()
        seq, seq_m = out[:, :idx + 1][:, :-1], tokens_masked[:, :idx + 1][:, 1:]
        ctx = context
        if ctx is not None and _get(dcfg, "context_emb_mode", "attention") == "cat":
            ctx = ctx[:, 1:]
        hidden, _ = tuple_transformer(sd, dec_prefix, dcfg, [seq, seq_m], causal=True,
                                      mask=torch.ones(seq.shape[:2], dtype=torch.bool), context=ctx, context_mask=context_mask,
                                      style=style[:, 1:], cached_decode=True)
        row = idx - 1
        if reference_hidden_row_defect:
            first_step = idx if first_step is None else first_step
            step = idx - first_step + 1                       # 1 = the cache-free first call
            t = 1
            while t * (t + 1) // 2 < step:
                t += 1
            row = first_step - 1 + (t - 1)
        lg = lm_head(sd, dec_prefix + "lm_head.", hidden[:, row], dcfg["token_embeddings"], keys,
                     _get(dcfg, "lm_head", {"_target_": "lm"}), only=dims)
        for d, (key, l) in zip(dims, lg.items()):
            l = l.clone()
            l[:, pad_token_id] = -float("inf")
            l[:, mask_token_id] = -float("inf")
            out[:, idx, d] = l.argmax(dim=-1)
    return out


# --------------------------------------------------------------------------------------
# A17  clip + AdamW                         experiments/optimizers.py:151-169
# --------------------------------------------------------------------------------------

def clip_adamw_step(params: List[Tensor], grads: List[Tensor], m: List[Tensor], v: List[Tensor], step: int, *,
                    lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6,
                    max_norm: Optional[float] = 2.0):
    """`clip_grad_norm_` (global L2, coef = max_norm/(norm+1e-6) clamped to 1) then torch.optim.AdamW."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = 1.0
    if max_norm is not None:
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for p, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        p.mul_(1 - lr * weight_decay)
        m_.mul_(b1).add_(g, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v_.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(m_, denom, value=-lr / (1 - b1 ** step))
    return total
