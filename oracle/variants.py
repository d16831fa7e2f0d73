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
