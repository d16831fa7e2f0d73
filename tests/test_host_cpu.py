"""CPU: host-side logic that mirrors the reference's contracts (constructor/registry errors, config container, synthetic
input contract, ALiBi slopes vs the reference fixture, weight-arena layout rules)."""
import os

import numpy as np
import pytest
import torch

from scoreperformer_amd.modules.constructor import Constructor, ModuleConfig, Registry, merge
from scoreperformer_amd.modules.transformer import ALiBiPositionalBias, AttentionConfig, Attention, TransformerRegistry
from scoreperformer_amd.synthetic import model_config, synthetic_batch, PERFORMANCE_VOCAB, PREDICTED_DIMS, IGNORED_DIMS
from scoreperformer_amd.utils.config import OmegaConf, DictConfig, MISSING

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_registry_and_constructor_error_contracts():
    reg = Registry()

    class A(torch.nn.Module, Constructor):
        def __init__(self, dim: int, depth: int = 2):
            super().__init__()
            self.dim, self.depth = dim, depth

    reg.register("a", A)
    with pytest.raises(KeyError, match="not found in registry"):      # constructor.py:120-124
        reg.get("missing")
    with pytest.raises(TypeError):
        reg.register(3)
    obj = A.init(OmegaConf.create({"dim": 8, "bogus": 1, "_target_": "a"}), depth=5)   # unknown keys dropped with a warning
    assert (obj.dim, obj.depth) == (8, 5)
    with pytest.raises(RuntimeError, match="mandatory"):              # constructor.py:61-63
        A.init({"dim": MISSING})
    assert reg.instantiate(OmegaConf.create({"_target_": "a", "dim": 3})).dim == 3
    assert set(TransformerRegistry.available_names) == {"default", "encoder", "decoder"}


def test_config_container_semantics():
    cfg = OmegaConf.create({"a": {"b": 1, "c": [1, 2]}, "d": None})
    assert cfg.a.b == 1 and cfg["a"]["c"] == [1, 2] and cfg.get("zzz", 7) == 7
    cfg.a.b = 5
    assert cfg["a"]["b"] == 5
    m = merge(cfg, {"a": {"e": 2}}, as_omega=True)
    assert m.a.b == 5 and m.a.e == 2
    assert isinstance(merge(AttentionConfig(dim=32), {"heads": 2})["dim"], int)


def test_alibi_slopes_match_reference_fixture():
    fix = dict(np.load(os.path.join(GOLD, "units.npz")))
    for h in (1, 2, 4, 6, 8, 12):
        np.testing.assert_allclose(ALiBiPositionalBias(h, h).slopes.view(-1).numpy(), fix[f"alibi/slopes/{h}"], rtol=1e-6)
    b = ALiBiPositionalBias(4, 4)
    np.testing.assert_array_equal(b.get_bias(5, 5, k=0).numpy(), fix["alibi/bias_5_5"])
    np.testing.assert_array_equal(b.get_bias(1, 7, k=6).numpy(), fix["alibi/bias_1_7"])


def test_synthetic_batch_follows_collator_contract():
    """MixedLMScorePerformanceCollator contract (collators/performance.py:239-255, score_performance.py:209-234)."""
    b = synthetic_batch(4, 64, seed=3, ragged=True)
    perf, mask, mp, lab = b["perf"], b["perf_mask"], b["masked_perf"], b["labels"]
    assert perf.shape == (4, 64, 12) and b["score"].shape == (4, 64, 10) and perf.dtype == torch.int64
    assert (perf[~mask] == 0).all() and (b["bars"][~mask] == 0).all()
    for k, v in enumerate(PERFORMANCE_VOCAB.values()):
        assert int(perf[..., k].max()) < v
    special = perf <= 3
    for d in PREDICTED_DIMS:
        assert (mp[..., d][~special[..., d]] == 1).all() and (mp[..., d][special[..., d]] == perf[..., d][special[..., d]]).all()
        assert (lab[..., d][~special[..., d]] == perf[..., d][~special[..., d]]).all() and (lab[..., d][special[..., d]] == -100).all()
    for d in IGNORED_DIMS:
        assert (mp[..., d] == perf[..., d]).all() and (lab[..., d] == -100).all()
    for seg in ("bars", "beats", "onsets"):
        s = b[seg]
        assert (s[:, 0] == 4).all()
        d = s[:, 1:] - s[:, :-1]
        assert ((d >= 0) | ~mask[:, 1:]).all()          # monotone on the valid prefix
    torch.testing.assert_close(synthetic_batch(4, 64, seed=3, ragged=True)["perf"], perf)   # seeded


def test_unsupported_variants_fail_loudly():
    with pytest.raises(NotImplementedError):
        Attention(dim=64, dim_head=128)                      # narrower heads run zero-padded (tests/test_attention_options_*.py)
    with pytest.raises(NotImplementedError):
        Attention(dim=64, max_attend=8)
    assert tuple(Attention(dim=64, dim_head=32, num_mem_kv=2).mem_k.shape) == (8, 2, 32)   # attention.py:98-101
    from scoreperformer_amd.models.scoreperformer.embeddings import TupleTokenEmbeddings
    with pytest.raises(ValueError):
        TupleTokenEmbeddings({"A": 8}, 8, mode="mean")
    with pytest.raises(AssertionError, match="should be the same for all keys"):      # embeddings.py:66-69
        TupleTokenEmbeddings({"A": 8, "B": 8}, {"A": 8, "B": 16}, mode="sum")
    te = TupleTokenEmbeddings({"A": 8, "B": 9}, 16, mode="sum", project_emb_dim=16)   # `sum`: one common width, no projection
    assert te.total_emb_dim == 16 and not hasattr(te, "project_emb")


def test_model_variants_construct_with_reference_layouts():
    from scoreperformer_amd.models import ScorePerformer, Performer
    m = ScorePerformer.init(model_config("tiny", lm_head="lm"))
    assert any(k.startswith("perf_decoder.model.lm_head.heads.Velocity") for k in m.state_dict())
    m2 = ScorePerformer.init(model_config("tiny", context_emb_mode="attention"))
    assert m2.perf_decoder.model.transformer.layer_types[:3] == ("a", "c", "f")
    cfg = model_config("tiny")
    p = Performer.init(OmegaConf.create({"transformer": dict(cfg["perf_decoder"], num_tokens=dict(PERFORMANCE_VOCAB), dim=128),
                                         "mode": "clm"}))
    assert type(p.transformer).__name__ == "ScorePerformerARWrapper"
    tied = m.score_encoder.token_emb.embs["Bar"] is m.perf_decoder.model.token_emb.embs["Bar"]
    assert tied


def test_small_zero_slices_do_not_share_a_version_counter():
    """Accumulators handed out of one pooled block are independent tensors to autograd: an in-place op on one (AccumulateGrad's `+=` on a
    bias gradient that came from the pool, when the reference's trainer accumulates over micro-batches) must not invalidate another one
    that an autograd node saved for its backward (found by tests/test_trainer_loop_gpu.py: 'modified by an inplace operation')."""
    import torch
    from scoreperformer_amd import ops
    a = ops.zeros_small(4, "cpu")
    b = ops.zeros_small(8, "cpu")
    assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()    # same block ...
    assert a.data_ptr() != b.data_ptr() and float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0

    class Keep(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, saved):
            ctx.save_for_backward(saved)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            (saved,) = ctx.saved_tensors          # raises if `saved`'s version moved since forward
            return g * 2 + saved.sum(), None

    x = torch.ones(3, requires_grad=True)
    y = Keep.apply(x, a)
    va = a._version
    b.add_(1.0)                                  # ... but an in-place op on the neighbour
    b.mul_(0.5)
    assert a._version == va                      # leaves this slice's version alone
    y.sum().backward()
    assert torch.equal(x.grad, torch.full((3,), 2.0))


def test_contiguous_note_spans_of_a_padding_mask():
    """Host logic of the decode engine's batch path (models/scoreperformer/wrappers.py): a padded batch goes through the engine when every
    sequence's notes are one contiguous block; holes or an empty row keep the module path."""
    import torch
    from scoreperformer_amd.models.scoreperformer.wrappers import contiguous_note_spans
    T, F = True, False
    m = torch.tensor([[T, T, T, T, T, T], [T, T, T, F, F, F], [F, F, T, T, T, T], [F, T, T, T, F, F], [F, F, F, F, F, T]])
    assert contiguous_note_spans(m) == [(0, 6), (0, 3), (2, 6), (1, 4), (5, 6)]
    assert contiguous_note_spans(torch.tensor([[T, T, F, T, T, T]])) is None            # a hole
    assert contiguous_note_spans(torch.tensor([[T, F, T, F, T, F]])) is None
    assert contiguous_note_spans(torch.tensor([[T, T, T], [F, F, F]])) is None          # an empty row
    assert contiguous_note_spans(torch.ones(2, 1, dtype=torch.bool)) == [(0, 1), (0, 1)]


def test_autocast_fence_is_transparent_without_autocast():
    """utils/amp.no_autocast: a plain call when autocast is off (the usual case: one flag read), name / docstring kept; under CPU autocast
    (the CUDA flag stays off) still a plain call."""
    import torch
    from scoreperformer_amd.utils.amp import no_autocast
    calls = []

    @no_autocast
    def f(x, *, k=1):
        """doc"""
        calls.append(torch.is_autocast_enabled("cuda"))
        return x * k
    assert f(3, k=2) == 6 and f.__name__ == "f" and f.__doc__ == "doc"
    with torch.autocast("cpu", dtype=torch.bfloat16):
        assert f(2) == 2
    assert calls == [False, False]


def test_latent_levels_fit_the_select_kernel_limits():
    """functional.latent_levels_fit: levels beyond the select kernel's LDS-resident bitmask / slot map keep the tensor-op path."""
    from scoreperformer_amd import functional as F_, ops
    assert F_.latent_levels_fit([(64, 1), (64, 155), (64, 1082)], 4096)
    assert not F_.latent_levels_fit([(64, ops.LATENT_SELECT_MAX_N // 64 + 1)], 4096)
    assert not F_.latent_levels_fit([(ops.LATENT_SELECT_MAX_B + 1, 4)], 4096)
    assert not F_.latent_levels_fit([(64, 100)], ops.LATENT_SELECT_MAX_K + 1)
