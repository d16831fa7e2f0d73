"""GPU: remaining API surface of the mirrored modules/models, each against fp32 PyTorch math or the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny(dev, **kw):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    cfg = model_config("tiny", **kw)
    model = ScorePerformer.init(model_config("tiny", **kw))
    sd = filled_state_dict(model, seed=2)
    model.load_state_dict(sd)
    arena = ParamArena(model, dev)
    return cfg, model, arena, sd


def test_evaluator_matches_torch_metrics(dev):
    from scoreperformer_amd.models import ScorePerformerEvaluator
    from scoreperformer_amd.synthetic import synthetic_batch
    cfg, model, arena, sd = _tiny(dev)
    model.train()
    batch = synthetic_batch(2, 48, seed=4, ragged=True, device=dev)
    out = model(**batch)
    ignore = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]
    tv = {k: torch.linspace(0, 1, v).tolist() for k, v in cfg["num_tokens"].items()}
    ev = ScorePerformerEvaluator(model, ignore_keys=ignore, weighted_distance=True, token_values=tv)
    metrics = ev(batch, out)
    labels = batch["labels"][:, 1:]
    keys = list(out.perf_decoder.logits.keys())
    preds = torch.stack([out.perf_decoder.logits[k].float().argmax(-1) for k in keys], -1)
    m = labels != -100
    assert abs(float(metrics["accuracy"]) - float((preds[m] == labels[m]).float().mean())) < 1e-6
    i = keys.index("Velocity")
    mi = labels[..., i] != -100
    assert abs(float(metrics["accuracy/Velocity"]) - float((preds[..., i][mi] == labels[..., i][mi]).float().mean())) < 1e-6
    assert "accuracy/Bar" not in metrics and "distance/Tempo" in metrics
    probs = out.perf_decoder.logits["Tempo"].float().softmax(-1)[labels[..., keys.index("Tempo")] != -100]
    assert torch.isfinite(metrics["distance/Tempo"]) and probs.shape[0] > 0


def test_untied_lm_head_variant_matches_oracle(dev):
    from oracle import ref_cpu
    from scoreperformer_amd.synthetic import synthetic_batch
    cfg, model, arena, sd = _tiny(dev, lm_head="lm")
    model.train()
    batch = synthetic_batch(2, 40, seed=6, ragged=True)
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    model.perf_encoder._z_override = [t.to(dev) for t in z]
    out = model(**{k: v.to(dev) for k, v in batch.items()})
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v) for k, v in sd.items()}
    ref = ref_cpu.score_performer_forward(sdg, cfg, batch, z, training=True)
    assert abs(float(out.loss) - float(ref["loss"])) <= 2e-2 * float(ref["loss"])
    arena.zero_grad()
    out.loss.backward()
    ref["loss"].backward()
    k = "perf_decoder.model.lm_head.heads.Velocity.weight"
    g = dict(model.named_parameters())[k].grad.float().cpu()
    assert (g - sdg[k].grad).abs().max() <= 0.06 * sdg[k].grad.abs().max() + 1e-5


def test_clm_wrapper_forward_and_generate(dev):
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import Performer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config, synthetic_batch, PERFORMANCE_VOCAB
    from scoreperformer_amd.utils.config import OmegaConf
    cfg = model_config("tiny", context_emb_mode="attention")
    dec = dict(cfg["perf_decoder"], num_tokens=dict(PERFORMANCE_VOCAB), dim=128, style_emb_dim=None, style_emb_mode="cat")
    dec["token_embeddings"] = dict(dec["token_embeddings"], _target_="simple")
    p = Performer.init(OmegaConf.create({"transformer": dec, "mode": "clm"}))
    ParamArena(p, dev)
    batch = synthetic_batch(2, 32, seed=8, device=dev)
    p.train()
    out = p(batch["perf"], mask=batch["perf_mask"], labels=batch["perf"].clone())
    assert torch.isfinite(out.loss) and len(out.losses) == 12
    out.loss.backward()
    p.eval()
    gen = p.transformer.generate(batch["perf"][:1, :4], seq_len=12, filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
    assert gen.shape[-1] == 12 and gen.shape[-2] >= 1
    assert int(gen.min()) >= 2        # PAD / MASK are banned (wrappers.py:257)


def test_mlm_single_run_unmask(dev):
    cfg, model, arena, sd = _tiny(dev)
    from scoreperformer_amd.models.scoreperformer.wrappers import ScorePerformerMLMWrapper
    from scoreperformer_amd.synthetic import synthetic_batch
    model.eval()
    batch = synthetic_batch(1, 24, seed=9, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    mlm = ScorePerformerMLMWrapper(model.perf_decoder.model)
    tokens = batch["masked_perf"].clone()
    with pytest.warns(UserWarning):
        out = mlm.unmask_tokens(tokens, single_run=True, x_extra=batch["masked_perf"], context=enc.score_embeddings,
                                style_embeddings=enc.perf_embeddings)
    assert int((out == 1).sum()) == 0 and (out[tokens != 1] == tokens[tokens != 1]).all()


def test_model_widths_off_the_8_element_grid(dev):
    """A model whose latent widths are not multiples of 8 (sum 12): the GEMM wrapper pads the operands; forward matches the oracle,
    backward runs and every gradient is finite."""
    from oracle import ref_cpu
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    vocab = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10, "PositionShift": 21,
             "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}
    kw = dict(preset="tiny", num_tokens=vocab, dim=32, heads=2, depths=(1, 1, 1), emb_dims=8, latent_dim=[4, 4, 2, 2], max_seq_len=64)
    cfg = model_config(**kw)
    model = ScorePerformer.init(model_config(**kw))
    sd = filled_state_dict(model, seed=1)
    model.load_state_dict(sd)
    arena = ParamArena(model, dev)
    model.train()
    batch = synthetic_batch(2, 24, seed=5, ragged=True, num_tokens=vocab)
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    model.perf_encoder._z_override = [t.to(dev) for t in z]
    out = model(**{k: v.to(dev) for k, v in batch.items()})
    arena.zero_grad()
    out.loss.backward()
    ref = ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True)
    assert abs(float(out.loss.detach()) - float(ref["loss"])) <= 1e-2 * float(ref["loss"])
    assert bool(torch.isfinite(arena.grads).all()) and float(arena.grads.abs().sum()) > 0


@pytest.mark.parametrize("name", ["no_cont_tokens", "no_masked_seq", "no_score_enc", "no_saln", "no_io_tie", "custom_hierarchy", "noisy_perf"])
def test_ablation_recipe_variants_match_the_reference_loss(dev, name):
    """The model variants of recipes/scoreperformer/ablation/*.yaml and custom_hierarchy.yaml: HIP forward against the reference's own
    loss (tests/golden/ablations.npz); backward runs."""
    from test_ablations_cpu import golden
    from oracle.variants import SMALL_VOCAB, ablation_config, variant_batch
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import synthetic_batch
    model = ScorePerformer.init(ablation_config(name))
    model.load_state_dict(filled_state_dict(model, seed=1))
    arena = ParamArena(model, dev)
    model.train()
    draws, loss, losses = golden(name)
    model.perf_encoder._z_override = [z.to(dev) for z in draws]
    batch = {k: v.to(dev) for k, v in variant_batch(name, synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)).items()}
    out = model(**batch)
    arena.zero_grad()
    out.loss.backward()
    assert abs(float(out.loss.detach()) - loss) < 1e-2 * abs(loss), (float(out.loss.detach()), loss)   # tiny model, bf16 GEMMs
    for k, v in losses.items():
        assert abs(float(out.losses[k]) - v) < 3e-2 * max(1.0, abs(v)), k
    assert bool(torch.isfinite(arena.grads).all())


def test_emb_dropout_and_post_act_ln_dropout_run_on_the_hip_path(dev):
    """`emb_dropout > 0` (models/scoreperformer/transformer.py:122,184: behind the concatenation, in front of the input projection) and
    `post_act_ln=True` with dropout (feedforward.py:56-59: the Dropout sits behind the LayerNorm): parity is statistical by construction
    (counter-based masks), so: eval mode equals the no-dropout model exactly, training mode differs, gradients flow, and with p -> 0
    the training loss meets the eval loss."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    from scoreperformer_amd.modules.transformer.feedforward import FeedForward
    torch.manual_seed(2)
    cfg = model_config("tiny")
    for k in ("score_encoder", "perf_encoder", "perf_decoder"):
        cfg[k]["emb_dropout"] = 0.3
    model = ScorePerformer.init(cfg)
    arena = ParamArena(model, dev)
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, seed=4, ragged=True).items()}
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)).to(dev) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    model.perf_encoder._z_override = z
    model.eval()
    with torch.no_grad():
        ev = float(model(**batch).loss)
    model.train()
    out = model(**batch)
    assert abs(float(out.loss) - ev) > 1e-4                      # the masks bite
    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    for m in model.modules():
        if hasattr(m, "emb_dropout") and hasattr(m.emb_dropout, "p"):
            m.emb_dropout.p = 1e-6
    assert abs(float(model(**batch).loss) - ev) < 5e-3
    # post-activation LayerNorm + dropout
    ff = FeedForward(dim=128, mult=2, glu=True, swish=True, post_act_ln=True, dropout=0.25).to(dev)
    x = torch.randn(2, 40, 128, device=dev).bfloat16().requires_grad_(True)
    ff.eval()
    y0 = ff(x)
    ff.train()
    y1 = ff(x)
    assert not torch.equal(y0, y1)
    y1.float().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
