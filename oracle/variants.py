"""Model configs of the reference's ablation recipes (recipes/scoreperformer/ablation/*.yaml, custom_hierarchy.yaml) at test size --
TEST INFRASTRUCTURE shared by the golden generator (oracle/refimport/make_golden_ablations.py) and the tests."""
from scoreperformer_amd.synthetic import model_config

SMALL_VOCAB = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10, "PositionShift": 21,
               "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}
NAMES = ["no_cont_tokens", "no_masked_seq", "no_score_enc", "no_saln", "no_io_tie", "custom_hierarchy", "noisy_perf"]


def ablation_config(name: str):
    base = lambda **kw: model_config(preset="tiny", num_tokens=SMALL_VOCAB, **kw)   # noqa: E731
    if name == "no_cont_tokens":        # ablation/no_cont_tokens.yaml: plain discrete embeddings instead of the value MLP
        c = base()
        for k in ("score_encoder", "perf_encoder", "perf_decoder"):
            c[k]["token_embeddings"].update(discrete=True, continuous=False, continuous_dense=False, discrete_ids=None)
        return c
    if name == "no_masked_seq":         # ablation/no_masked_seq.yaml: the decoder does not see the masked copy of the next note
        c = base()
        te = c["perf_decoder"]["token_embeddings"]
        te["_target_"] = "simple"
        te.pop("multiseq_mode", None)
        return c
    if name == "no_score_enc":          # ablation/no_score_enc.yaml
        c = base()
        c["score_encoder"] = None
        return c
    if name == "no_saln":               # ablation/no_saln.yaml: style embeddings concatenated instead of adaptive LayerNorm
        return base(style_emb_mode="cat")
    if name == "no_io_tie":             # ablation/no_io_tie.yaml: untied LM head
        return base(lm_head="lm")
    if name == "custom_hierarchy":      # custom_hierarchy.yaml: a subset of the latent hierarchy
        c = base()
        c["perf_encoder"]["aggregate_mode"] = ["mean", "bar_mean"]
        c["perf_encoder"]["latent_dim"] = [16, 16]
        return c
    if name == "noisy_perf":            # base model; the batch carries a noisy performance for the style encoder (base.yaml:50)
        return base()
    raise KeyError(name)


def variant_batch(name: str, batch):
    """Inputs of a variant: `noisy_perf` adds a perturbed copy of the performance (velocity / timing dims re-drawn)."""
    if name != "noisy_perf":
        return batch
    import torch
    g = torch.Generator().manual_seed(99)
    noisy = batch["perf"].clone()
    for dim, key in ((3, "Velocity"), (10, "RelOnsetDev"), (11, "RelPerfDuration")):
        draw = torch.randint(4, SMALL_VOCAB[key], noisy[..., dim].shape, generator=g)
        noisy[..., dim] = torch.where(batch["perf_mask"] & (noisy[..., dim] > 3), draw, noisy[..., dim])
    return dict(batch, noisy_perf=noisy, noisy_perf_mask=batch["perf_mask"].clone())


# ---- second variant set (round 2): code paths no shipped recipe switches on, pinned to the reference all the same ------------------
# lm-tied-split head (models/scoreperformer/embeddings.py:364-390), absolute positional embeddings (transformer.py:124-125,168-169),
# GELU and un-gated feed-forwards (feedforward.py:35-64), the regression head and its L1 loss (embeddings.py:400-420,
# wrappers.py:61-78), and the decoder-only `Performer` in CLM / MLM / MixedLM mode (model.py:62-122, wrappers.py:290-307).
NAMES2 = ["lm_tied_split", "abs_pos_emb", "ff_gelu", "ff_gelu_glu", "ff_silu_plain", "regression_head",
          "performer_clm", "performer_mlm", "performer_mixlm"]


def variant2_config(name: str):
    base = lambda **kw: model_config(preset="tiny", num_tokens=SMALL_VOCAB, **kw)   # noqa: E731
    if name == "lm_tied_split":
        return base(lm_head="lm-tied-split")
    if name == "abs_pos_emb":
        c = base()
        for k in ("score_encoder", "perf_encoder", "perf_decoder"):
            c[k]["use_abs_pos_emb"] = True
        return c
    if name in ("ff_gelu", "ff_gelu_glu", "ff_silu_plain"):
        c = base()
        glu, swish = {"ff_gelu": (False, False), "ff_gelu_glu": (True, False), "ff_silu_plain": (False, True)}[name]
        for k in ("score_encoder", "perf_encoder", "perf_decoder"):
            c[k]["transformer"]["feed_forward"].update(glu=glu, swish=swish)
        return c
    if name == "regression_head":
        c = base()
        c["perf_decoder"]["regression_head"] = dict(regression_keys=["Velocity", "Tempo", "RelOnsetDev"])
        return c
    if name.startswith("performer_"):
        mode = name.split("_", 1)[1]
        full = base(lm_head="lm-tied" if mode != "mlm" else "lm")
        tr = full["perf_decoder"]
        tr["context_emb_mode"], tr["style_emb_mode"], tr["style_emb_dim"] = "attention", "cat", None   # no encoders: nothing to condition on
        tr["num_tokens"], tr["dim"] = dict(SMALL_VOCAB), full["dim"]
        if mode != "mixlm":
            te = tr["token_embeddings"]
            te["_target_"] = "simple"
            te.pop("multiseq_mode", None)
        if mode == "mlm":
            tr["transformer"]["_target_"] = "encoder"
        from scoreperformer_amd.utils.config import OmegaConf
        return OmegaConf.create(dict(transformer=tr, mode=mode))
    raise KeyError(name)


def variant2_inputs(name: str, batch):
    """Forward kwargs of a variant: the full MixedLM batch for ScorePerformer models, (perf, mask, labels[, masked_perf]) for Performer."""
    if not name.startswith("performer_"):
        return batch
    mode = name.split("_", 1)[1]
    if mode == "mlm":    # masked-LM: the model reads the masked performance and predicts the masked dims in place
        return dict(perf=batch["masked_perf"], mask=batch["perf_mask"], labels=batch["labels"])
    out = dict(perf=batch["perf"], mask=batch["perf_mask"], labels=batch["labels"])
    if mode == "mixlm":
        out["masked_perf"] = batch["masked_perf"]
    return out


# ---- third variant set (round 4): the embedding modes that are the reference's DEFAULTS but that no shipped recipe keeps --------------
# `multiseq_mode="pre-sum"` (embeddings.py:171,231-241: the decoder's two sequences are summed per key before norm / projection) and
# `TupleTokenEmbeddings(mode="sum")` (embeddings.py:66-69,117,141: per-key embeddings of one common width summed, only normalised);
# plus two style-encoder variants: `hierarchical_with_context=False` and the per-note `aggregate_mode="same"`.
NAMES3 = ["multiseq_pre_sum", "emb_mode_sum", "hier_no_context", "agg_same", "isolated_bar"]


def variant3_config(name: str):
    base = lambda **kw: model_config(preset="tiny", num_tokens=SMALL_VOCAB, **kw)   # noqa: E731
    if name == "multiseq_pre_sum":
        c = base()
        c["perf_decoder"]["token_embeddings"]["multiseq_mode"] = "pre-sum"
        return c
    if name == "emb_mode_sum":          # width = model width: `sum` mode never projects (embeddings.py:141); the tied head needs the
        c = base(lm_head="lm")          # concatenated layout, so the untied head
        for k in ("score_encoder", "perf_encoder", "perf_decoder"):
            c[k]["token_embeddings"].update(mode="sum", emb_dims=int(c["dim"]))
        c["perf_decoder"]["token_embeddings"]["multiseq_mode"] = "pre-sum"
        return c
    if name == "hier_no_context":       # mmd_transformer.py:152-156,255-262: level i reads ONLY level i - 1's embeddings (no hidden states)
        c = base()
        c["perf_encoder"]["hierarchical_with_context"] = False
        return c
    if name == "agg_same":              # mmd_transformer.py:19,343-344: one latent per NOTE (no aggregation) as the last level
        c = base()
        c["perf_encoder"]["aggregate_mode"] = ["mean", "bar_mean", "same"]
        c["perf_encoder"]["latent_dim"] = [16, 8, 8]
        c["perf_encoder"]["latent_dropout"] = [0.0, 0.0, 0.0]
        c["perf_decoder"]["style_emb_dim"] = 32
        return c
    if name == "isolated_bar":          # mmd_transformer.py:186-200: the style encoder reads bar tokens masked; its block-diagonal attention
        c = base()                      # mask is built and then dropped by TupleTransformer.forward (never reaches the layer stack)
        c["perf_encoder"]["aggregate_mode"] = ["isolated_bar_mean", "beat_mean"]
        c["perf_encoder"]["latent_dim"] = [16, 8]
        c["perf_encoder"]["latent_dropout"] = [0.0, 0.0]
        c["perf_decoder"]["style_emb_dim"] = 24
        return c
    raise KeyError(name)
