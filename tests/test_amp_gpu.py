"""GPU: `torch.cuda.amp.autocast` around the public entry points changes NOTHING (recipes/default.yaml:89 -> trainer.py:449).

The precision of this path is fixed by its kernels; every public forward fences autocast off (`scoreperformer_amd/utils/amp.py`).  These
tests run the same call with and without an enabled fp16 autocast context and require the same results: the train forward / backward
of both bindings, the module-level operators a user may call on their own, the evaluator, and the cached greedy decode on both the
engine and the module path.  "The same" is bit-identical wherever the kernels are run-to-run deterministic (logits, tokens, module
outputs) and 3e-6 relative where sums go through float atomics (loss entries, gradients: two plain runs differ by 1-3e-7 there,
`tools/det_probe.py`) -- three orders of magnitude below what a recast to fp16 of any intermediate would leave."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SMALL_VOCAB = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10,
               "PositionShift": 21, "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}


def _autocast(on):
    from torch.cuda import amp
    return amp.autocast(enabled=on)


def _tiny(dev, arena, dropout=0.0):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    model = ScorePerformer.init(model_config("tiny", dropout=dropout))
    model.load_state_dict(filled_state_dict(model, seed=11))
    if arena:
        model._test_arena = ParamArena(model, dev)
    else:
        model.to(dev)
    return model


@pytest.mark.parametrize("arena", [False, True], ids=["zero-edit", "arena"])
def test_train_forward_backward_is_bit_identical_under_autocast(dev, arena):
    """Loss, every entry of `losses`, the decoder's logits and every parameter gradient: the same with autocast on and off, on both
    bindings (logits bit for bit; sums through float atomics to 3e-6).  The backward runs outside the context, as optimizers.py:152 does."""
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config("tiny", dropout=0.0)
    model = _tiny(dev, arena)
    model.train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, seed=3, ragged=True).items()}
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)).to(dev) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    runs = []
    for on in (False, True):
        if arena:
            model._test_arena.zero_grad()
        else:
            model.zero_grad(set_to_none=True)
        model.perf_encoder._z_override = z
        with _autocast(on):
            assert torch.is_autocast_enabled("cuda") == on
            out = model(**batch)
        (out.loss * 65536.0).backward()                             # the scaler's factor (optimizers.py:152)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        runs.append((out.loss.detach().clone(), {k: v.detach().clone() for k, v in out.losses.items()},
                     {k: v.detach().clone() for k, v in out.perf_decoder.logits.items()}, grads))
    (l0, ls0, lg0, g0), (l1, ls1, lg1, g1) = runs
    assert l0.dtype == l1.dtype == torch.float32 and abs(float(l0) - float(l1)) <= 3e-6 * abs(float(l0))
    assert ls0.keys() == ls1.keys()
    assert all(ls0[k].dtype == ls1[k].dtype and abs(float(ls0[k]) - float(ls1[k])) <= 3e-6 * max(1.0, abs(float(ls0[k]))) for k in ls0)
    assert all(lg0[k].dtype == lg1[k].dtype and torch.equal(lg0[k], lg1[k]) for k in lg0)          # deterministic kernels: bit for bit
    assert g0.keys() == g1.keys() and len(g0) > 50
    assert all(g0[k].dtype == g1[k].dtype == torch.float32 for k in g0)
    worst = max(float((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-30)) for k in g0)
    assert worst <= 3e-6, worst
    assert all(torch.isfinite(v).all() for v in g0.values())


def test_loss_scale_is_linear_through_the_backward(dev):
    """GradScaler's premise on this path: gradients of 65 536 x loss, divided by 65 536 (`unscale_`), equal the gradients of the loss to
    bf16 rounding -- a power-of-two factor commutes with every rounding in the backward unless a value leaves the exponent range, which
    would show as inf / nan or as a flushed (zero) entry.  Bound: whole-vector relative L2 <= 1e-6, i.e. exact up to fp32 summation
    order inside the kernels' atomics."""
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config("tiny", dropout=0.0)
    model = _tiny(dev, arena=False)
    model.train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, seed=4, ragged=True).items()}
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)).to(dev) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    vecs = []
    for scale in (1.0, 65536.0):
        model.zero_grad(set_to_none=True)
        model.perf_encoder._z_override = z
        (model(**batch).loss * scale).backward()
        vecs.append(torch.cat([p.grad.double().reshape(-1) / scale for p in model.parameters() if p.grad is not None]))
    assert torch.isfinite(vecs[1]).all()
    rel = float((vecs[0] - vecs[1]).norm() / vecs[0].norm())
    assert rel <= 1e-6, rel
    assert int((vecs[0] != 0).sum()) == int((vecs[1] != 0).sum())    # nothing flushed, nothing created


def test_module_level_operators_under_autocast(dev):
    """The operators of `scoreperformer.modules` called on their own inside an autocast region (a user's own model code): same bits
    as outside, same output dtypes."""
    from scoreperformer_amd.modules import AdaptiveLayerNorm
    from scoreperformer_amd.modules.transformer import Attention, FeedForward, Decoder
    torch.manual_seed(0)
    x = torch.randn(2, 48, 128, device=dev)
    cond = torch.randn(2, 48, 16, device=dev)
    mask = torch.ones(2, 48, dtype=torch.bool, device=dev)
    mask[1, 40:] = False
    mods = {
        "attention": (Attention(dim=128, heads=2, dim_head=64, causal=True, one_kv_head=True, alibi_pos_bias=True, alibi_learned=True).to(dev),
                      lambda m: m(x, mask=mask)[0]),
        "feed_forward": (FeedForward(dim=128, mult=4, glu=True, swish=True).to(dev), lambda m: m(x)),
        "ada_layer_norm": (AdaptiveLayerNorm(128, 16).to(dev), lambda m: m(x, condition=cond)),
        "decoder": (Decoder(dim=128, depth=2, heads=2, attention=dict(dim_head=64, one_kv_head=True, alibi_pos_bias=True, alibi_learned=True),
                            feed_forward=dict(mult=4, glu=True, swish=True)).to(dev), lambda m: m(x, mask=mask)),
    }
    for name, (mod, call) in mods.items():
        mod.eval()
        outs = []
        for on in (False, True):
            with torch.no_grad(), _autocast(on):
                y = call(mod)
            outs.append(y[0] if isinstance(y, tuple) else y)
        assert outs[0].dtype == outs[1].dtype, name
        assert torch.equal(outs[0], outs[1]), name


def test_greedy_decode_and_evaluator_under_autocast(dev):
    """`unmask_tokens` (engine and module path) and the per-step evaluator inside an autocast region: the reference's fixture tokens bit
    for bit on the engine, the same tokens as without autocast on the module path, identical metrics."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer, ScorePerformerEvaluator
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy.npz"), allow_pickle=False))
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB))
    model.load_state_dict(filled_state_dict(model, seed=3))
    ParamArena(model, dev)
    model.eval()
    tokens = torch.from_numpy(fix["in/tokens"]).to(dev)
    masked = torch.from_numpy(fix["in/masked_perf"]).to(dev)
    ctx = torch.from_numpy(fix["out/score_embeddings"]).to(dev)
    sty = torch.from_numpy(fix["out/perf_embeddings"]).to(dev)
    dec = model.perf_decoder
    for engine in (True, False):
        dec.use_decode_engine = engine
        outs = []
        for on in (False, True):
            with _autocast(on):
                outs.append(dec.unmask_tokens(tokens, masked, context=ctx, style_embeddings=sty, filter_logits_fn=top_k,
                                              filter_kwargs={"k": 1}, disable_tqdm=True))
        assert torch.equal(outs[0], outs[1]), engine
        if engine:
            assert int((outs[1].cpu().numpy() != fix["out/tokens"]).sum()) == 0

    from scoreperformer_amd.synthetic import model_config as mc
    model2 = _tiny(dev, arena=True)
    model2.train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, seed=9, ragged=True).items()}
    ignore = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]
    tv = {k: (torch.linspace(0, 1, v) ** 2 * 7).tolist() for k, v in mc("tiny")["num_tokens"].items()}
    for attach in (False, True):
        evaluator = ScorePerformerEvaluator(model2, ignore_keys=ignore, weighted_distance=True, token_values=tv)
        if attach:
            evaluator.attach()
        metrics = []
        for on in (False, True):
            torch.manual_seed(1)
            with torch.no_grad(), _autocast(on):
                out = model2(**batch)
                metrics.append(evaluator(batch, out))
        evaluator.detach()
        assert metrics[0].keys() == metrics[1].keys() and len(metrics[0]) > 0
        assert all(metrics[0][k].dtype == metrics[1][k].dtype and abs(float(metrics[0][k]) - float(metrics[1][k])) <= 1e-5 * max(1.0, abs(float(metrics[0][k])))
                   for k in metrics[0]), attach          # (the fused metric sums go through float atomics)
