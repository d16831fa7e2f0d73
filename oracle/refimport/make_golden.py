"""Generate tests/golden/*.npz by running the REFERENCE (``/root/reference``) on CPU fp32.

Authoring-container only (the reference does not travel to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden

Fixtures hold inputs and reference outputs only; weights are regenerated in the
tests from ``oracle.weights`` (pure function of key name / shape / seed).
No reference source is copied; it is imported from where it lies.
"""
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
from oracle.refimport import stubs  # noqa: E402

stubs.install()
warnings.filterwarnings("ignore")

from scoreperformer.models import ScorePerformer  # noqa: E402  (the reference)
from scoreperformer.modules.sampling import top_k  # noqa: E402

from oracle.weights import filled_state_dict  # noqa: E402
from scoreperformer_amd.synthetic import model_config, synthetic_batch, PREDICTED_DIMS  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")

# small vocabularies keep the fixtures small; sizes are constructor inputs
SMALL_VOCAB = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29,
               "TimeSig": 10, "PositionShift": 21, "NotesInOnset": 16, "PositionInOnset": 16,
               "RelOnsetDev": 45, "RelPerfDuration": 25}

VARIANTS = {
    # name: (config kwargs, batch kwargs)
    "tiny_mixlm": (dict(preset="tiny"), dict(batch=3, seq_len=48, ragged=True, seed=11)),
    "tiny_xattn_mha": (dict(preset="tiny", context_emb_mode="attention", style_emb_mode="cat", one_kv_head=False,
                            alibi_learned=False), dict(batch=2, seq_len=40, ragged=True, seed=12)),
    "tiny_full_vocab": (dict(preset="tiny", num_tokens=None), dict(batch=2, seq_len=32, ragged=False, seed=13)),
}


class RandnRecorder:
    """Records the z ~ N(0, I) draws of `MMDLoss.forward` (mmd_transformer.py:519)."""

    def __init__(self):
        self.samples = []
        self._orig = torch.randn

    def __enter__(self):
        def randn(*size, **kw):
            out = self._orig(*size, **kw)
            self.samples.append(out.clone())
            return out

        torch.randn = randn
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


def build(cfg_kwargs, seed=0):
    kw = dict(cfg_kwargs)
    if "num_tokens" not in kw:
        kw["num_tokens"] = SMALL_VOCAB
    elif kw["num_tokens"] is None:
        kw.pop("num_tokens")
    cfg = model_config(**kw)
    model = ScorePerformer.init(model_config(**kw))
    model.load_state_dict(filled_state_dict(model, seed=seed), strict=True)
    return cfg, model, kw


def run_train_variant(name, cfg_kwargs, batch_kwargs):
    cfg, model, kw = build(cfg_kwargs)
    vocab = kw.get("num_tokens")
    batch = synthetic_batch(num_tokens=vocab, **batch_kwargs)
    if batch["perf"].shape[0] >= 3:
        batch["deadpan_mask"][1] = True
    model.train()
    torch.manual_seed(1234)
    with RandnRecorder() as rec:
        out = model(**batch)
    out.loss.backward()
    named = dict(model.named_parameters())
    grads = {k: p.grad for k, p in named.items() if p.grad is not None}
    fix = {f"in/{k}": v.numpy() for k, v in batch.items()}
    sd = model.state_dict()
    fix["meta/state_dict_keys"] = np.array("\n".join(sd.keys()))
    fix["meta/state_dict_shapes"] = np.array("\n".join(",".join(map(str, v.shape)) for v in sd.values()))
    fix["meta/num_params"] = np.array(sum(p.numel() for p in model.parameters()))
    for i, z in enumerate(rec.samples):
        fix[f"z/{i}"] = z.numpy()
    fix["out/loss"] = out.loss.detach().numpy()
    for k, v in out.losses.items():
        fix[f"losses/{k}"] = v.detach().numpy()
    for i, (k, lg) in enumerate(out.perf_decoder.logits.items()):
        if i in PREDICTED_DIMS:
            fix[f"logits/{k}"] = lg.detach().numpy()
        fix[f"logits_sum/{k}"] = lg.detach().double().sum().numpy()
    fix["out/hidden_state"] = out.perf_decoder.hidden_state.detach().numpy()
    fix["out/perf_embeddings"] = out.perf_encoder.embeddings.detach().numpy()
    fix["out/score_embeddings"] = out.score_encoder.hidden_state.detach().numpy()
    for i, lat in enumerate(out.perf_encoder.latents):
        fix[f"latents/{i}"] = lat.detach().numpy()
    for k, g in grads.items():
        fix[f"gradnorm/{k}"] = g.double().norm().numpy()
    # a few full gradients (small tensors, and the tied ones that collect 4 contributions)
    for k in grads:
        if any(t in k for t in ("learned_logslopes", "embs.Velocity.index_weight", "vae_head.bar_mean.linear.weight",
                                "layers.0.0.0.linear.bias", "perf_decoder.model.transformer.layers.1.1.ff.3.weight", "embs.Tempo.value_layer.1.0.weight",
                                "perf_decoder.model.token_emb.norm.weight")):
            fix[f"grad/{k}"] = grads[k].clone().numpy()

    # one optimizer step: clip_grad_norm_(2.0) + AdamW(lr 2e-4, wd 1e-6) as experiments/optimizers.py:151-169
    params = [p for p in model.parameters() if p.grad is not None]
    total_norm = torch.nn.utils.clip_grad_norm_(params, 2.0)
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-6)
    before = {k: p.detach().clone() for k, p in named.items()}
    opt.step()
    fix["opt/total_norm"] = total_norm.numpy()
    for k in ("perf_decoder.model.transformer.layers.0.1.to_q.weight", "perf_encoder.vae_head.mean.linear.bias"):
        if k in named:
            fix[f"opt/delta/{k}"] = (named[k].detach() - before[k]).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fix)
    print(name, "loss", float(out.loss), {k: round(float(v), 5) for k, v in out.losses.items()},
          "n_arrays", len(fix))


def run_greedy(name="tiny_greedy"):
    """Cached greedy `unmask_tokens` (wrappers.py:325-407) on a 40-note window, eval mode."""
    cfg, model, kw = build(dict(preset="tiny"), seed=3)
    model.eval()
    batch = synthetic_batch(1, 40, num_tokens=kw["num_tokens"], seed=21)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"],
                                     score_mask=batch["score_mask"], bars=batch["bars"], beats=batch["beats"],
                                     onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"], compute_loss=False)
    tokens = batch["masked_perf"].clone()  # known score dims, MASK on the 4 predicted dims
    tokens[:, 0] = batch["perf"][:, 0]
    out, caches = model.perf_decoder.unmask_tokens(
        tokens, batch["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
        filter_logits_fn=top_k, filter_kwargs={"k": 1}, return_caches=True, disable_tqdm=True)
    fix = {f"in/{k}": v.numpy() for k, v in batch.items()}
    fix["in/tokens"] = tokens.numpy()
    fix["out/tokens"] = out.numpy()
    fix["out/score_embeddings"] = enc.score_embeddings.numpy()
    fix["out/perf_embeddings"] = enc.perf_embeddings.numpy()
    fix["cache/token_emb_shape"] = np.array(caches.token_emb.shape)
    fix["cache/n_hiddens"] = np.array(len(caches.transformer.hiddens))
    fix["cache/keys0_shape"] = np.array(caches.transformer.attention[0].keys.shape)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fix)
    print(name, "tokens", out.shape, "changed", int((out != tokens).sum()))


def run_units(name="units"):
    """Module-level vectors from the reference classes."""
    from scoreperformer.modules.transformer import Attention, FeedForward, ALiBiPositionalBias
    from scoreperformer.modules.layers import AdaptiveLayerNorm
    from scoreperformer.models.scoreperformer.mmd_transformer import MMDLoss
    fix = {}
    g = torch.Generator().manual_seed(5)
    for h in (1, 2, 4, 6, 8, 12):
        fix[f"alibi/slopes/{h}"] = ALiBiPositionalBias(h, h).slopes.view(-1).numpy()
    ab = ALiBiPositionalBias(4, 4)
    fix["alibi/bias_5_5"] = ab.get_bias(5, 5, k=0).numpy()
    fix["alibi/bias_1_7"] = ab.get_bias(1, 7, k=6).numpy()
    x = torch.randn(2, 9, 32, generator=g)
    mask = torch.ones(2, 9, dtype=torch.bool)
    mask[1, 6:] = False
    fix["attn/x"], fix["attn/mask"] = x.numpy(), mask.numpy()
    for causal in (False, True):
        for mqa in (False, True):
            for learned in (False, True):
                att = Attention(dim=32, dim_head=8, heads=4, causal=causal, one_kv_head=mqa, alibi_pos_bias=True,
                                alibi_learned=learned).eval()
                tag = f"attn/c{int(causal)}_m{int(mqa)}_l{int(learned)}"
                att.load_state_dict(filled_state_dict(att, seed=7))
                with torch.no_grad():
                    out, inter, _ = att(x, mask=mask)
                fix[tag] = out.numpy()
    for glu in (False, True):
        for swish in (False, True):
            ff = FeedForward(dim=32, mult=2, glu=glu, swish=swish).eval()
            ff.load_state_dict(filled_state_dict(ff, seed=8))
            with torch.no_grad():
                fix[f"ff/g{int(glu)}_s{int(swish)}"] = ff(x).numpy()
    ada = AdaptiveLayerNorm(32, 6)
    ada.load_state_dict(filled_state_dict(ada, seed=9))
    cond = torch.randn(2, 9, 6, generator=g)
    fix["ada/cond"] = cond.numpy()
    with torch.no_grad():
        fix["ada/out"] = ada(x, condition=cond).numpy()
    z, y = torch.randn(256, 8, generator=g), 0.7 * torch.randn(77, 8, generator=g) + 0.2
    fix["mmd/z"], fix["mmd/y"] = z.numpy(), y.numpy()
    fix["mmd/out"] = MMDLoss.compute_mmd(z, y).numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fix)
    print(name, len(fix), "arrays")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    run_units()
    for name, (ck, bk) in VARIANTS.items():
        run_train_variant(name, ck, bk)
    run_greedy()


if __name__ == "__main__":
    main()
