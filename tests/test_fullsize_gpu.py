"""GPU: the train step at BASELINE config 3's FULL size (64 sequences x 2048 notes, the C3 model) through size-independent properties
-- the CPU oracle cannot run this size in test time, so the checks are the ones the domain offers:

  * LINEARITY IN THE BATCH / ORDER INDEPENDENCE.  No op of the model mixes samples (LayerNorm, attention, segment means are per sample;
    the MMD term is the one exception and is left out), so a batch of 32 sequences followed by the same 32 in reversed order has twice
    the per-key cross-entropy SUMS and label COUNTS of the 32 alone, the same mean cross-entropy and the same gradient.  A kernel whose
    result depended on the row count of a launch or on where in the batch a sequence sits (tile edges, split-K plans, persistent walks,
    XCD maps, the attention block order all change between 32 and 64 sequences) breaks this.
  * PADDING INVARIANCE.  Appending masked notes behind every sequence changes neither sums, counts nor gradients.

Reference: models/scoreperformer/model.py:280-341 (forward), wrappers.py:49-59,409-431 (per-key mean cross-entropy)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, seq):
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    torch.manual_seed(1234)
    model = ScorePerformer.init(model_config("c3", max_seq_len=seq, dropout=0.0))
    arena = ParamArena(model, dev)
    model.train()
    return model, arena


def _run(model, arena, batch):
    out = model(**batch)
    ce = out.loss - out.perf_encoder.loss            # the cross-entropy part: mean over the predicted keys of sum / count
    arena.zero_grad()
    ce.backward()
    torch.cuda.synchronize()
    sums = {k: (float(v[0].detach()), float(v[1].detach())) for k, v in out.perf_decoder.ce_sums.items()}
    sums = {k: v for k, v in sums.items() if v[1] > 0}            # the predicted keys (the others carry no labels: count 0)
    assert len(sums) == 4
    return float(ce.detach()), sums, arena.grads.clone()


def _check_grad(got, want, tag):
    err = float((got - want).norm() / want.norm())
    assert err <= 2e-4, (tag, err)                      # bf16 operands are identical; only fp32 summation orders differ (measured 1.4e-5)
    return err


def test_c3_full_size_step_is_linear_in_the_batch(dev):
    """64 sequences = 32 distinct ones + the same 32 in reversed order, against the 32 alone.  Sums and counts double, the mean
    cross-entropy is the same, and because every backward value of the full batch is EXACTLY half the half batch's (1 / count is
    halved: a power of two, so every bf16 rounding falls the same way) the gradients agree to fp32 summation order -- whereas two
    halves with different label counts only agree to the bf16 noise of two independent roundings (~5 %, measured)."""
    from scoreperformer_amd.synthetic import synthetic_batch
    model, arena = _setup(dev, 2048)
    half = synthetic_batch(32, 2048, seed=5, ragged=True, device=dev)
    full = {k: torch.cat([v, v.flip(0)], dim=0) for k, v in half.items()}
    ce, sums, grad = _run(model, arena, full)
    ce_h, sums_h, grad_h = _run(model, arena, half)
    for k, (s, c) in sums.items():
        assert c == 2 * sums_h[k][1], (k, c, sums_h[k][1])
        assert abs(s - 2 * sums_h[k][0]) <= 2e-5 * abs(s), (k, s, sums_h[k][0])
    assert abs(ce - ce_h) <= 2e-5 * ce
    err = _check_grad(grad, grad_h, "32 + 32 reversed vs 32")
    print(f"full size (64 x 2048): CE {ce:.6f} vs {ce_h:.6f} for the 32 distinct sequences alone; gradient relative L2 difference {err:.2e}")


def test_c3_full_size_step_ignores_padding(dev):
    from scoreperformer_amd.synthetic import synthetic_batch
    model, arena = _setup(dev, 2048)
    short = synthetic_batch(64, 1792, seed=6, ragged=True, device=dev)
    ce, sums, grad = _run(model, arena, short)
    padded = {}
    for k, v in short.items():
        if v.ndim == 1:
            padded[k] = v
        else:
            pad = torch.zeros((v.shape[0], 256) + tuple(v.shape[2:]), dtype=v.dtype, device=dev)
            if k == "labels":
                pad.fill_(-100)
            padded[k] = torch.cat([v, pad], dim=1)
    ce2, sums2, grad2 = _run(model, arena, padded)
    for k, (s, c) in sums.items():
        assert sums2[k][1] == c and abs(sums2[k][0] - s) <= 2e-5 * abs(s), (k, sums[k], sums2[k])
    assert abs(ce - ce2) <= 2e-5 * ce
    err = _check_grad(grad2, grad, "padding")
    print(f"full size: 256 masked notes appended to every sequence: CE {ce:.6f} vs {ce2:.6f}, gradient relative L2 difference {err:.2e}")
