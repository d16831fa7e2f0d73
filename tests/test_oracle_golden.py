"""CPU: pin the oracle (oracle/ref_cpu.py) against fixtures produced by the reference itself, and check that the
product's module tree has the reference's state_dict layout (checkpoint compatibility)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.weights import filled_state_dict
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SMALL_VOCAB = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10,
               "PositionShift": 21, "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}
VARIANTS = {
    "tiny_mixlm": dict(preset="tiny", num_tokens=SMALL_VOCAB),
    "tiny_xattn_mha": dict(preset="tiny", context_emb_mode="attention", style_emb_mode="cat", one_kv_head=False,
                           alibi_learned=False, num_tokens=SMALL_VOCAB),
    "tiny_full_vocab": dict(preset="tiny"),
}


def load(name):
    return dict(np.load(os.path.join(GOLD, f"{name}.npz"), allow_pickle=False))


def build(name, seed=0):
    cfg = model_config(**VARIANTS[name])
    model = ScorePerformer.init(model_config(**VARIANTS[name]))
    sd = filled_state_dict(model, seed=seed)
    return cfg, model, sd


@pytest.mark.parametrize("name", list(VARIANTS))
def test_state_dict_layout_matches_reference(name):
    fix = load(name)
    _, model, _ = build(name)
    sd = model.state_dict()
    ref_keys = str(fix["meta/state_dict_keys"]).split("\n")
    ref_shapes = str(fix["meta/state_dict_shapes"]).split("\n")
    assert list(sd.keys()) == ref_keys
    assert [",".join(map(str, v.shape)) for v in sd.values()] == ref_shapes
    assert sum(p.numel() for p in model.parameters()) == int(fix["meta/num_params"])


@pytest.mark.parametrize("name", list(VARIANTS))
def test_oracle_matches_reference_forward_backward(name):
    fix = load(name)
    cfg, _, sd = build(name)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v)
          for k, v in sd.items()}
    # tied tensors share one leaf so that gradients accumulate like in the reference module tree
    from oracle.weights import canonical
    leaves = {}
    for k in list(sd):
        c = canonical(k)
        if c in leaves:
            sd[k] = leaves[c]
        else:
            leaves[c] = sd[k]
    batch = {k[3:]: torch.from_numpy(v) for k, v in fix.items() if k.startswith("in/")}
    z = [torch.from_numpy(fix[f"z/{i}"]) for i in range(4)]
    out = ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True)
    assert abs(float(out["loss"]) - float(fix["out/loss"])) < 2e-5
    ref_losses = {k[7:]: float(v) for k, v in fix.items() if k.startswith("losses/")}
    assert set(out["losses"]) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(float(out["losses"][k]) - v) < 2e-5, k
    for k, v in fix.items():
        if k.startswith("logits/"):
            np.testing.assert_allclose(out["logits"][k[7:]].detach().numpy(), v, atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(out["hidden_state"].detach().numpy(), fix["out/hidden_state"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(out["perf_embeddings"].detach().numpy(), fix["out/perf_embeddings"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(out["score_embeddings"].detach().numpy(), fix["out/score_embeddings"], atol=3e-5, rtol=1e-4)
    out["loss"].backward()
    checked = 0
    for k, v in fix.items():
        if k.startswith("gradnorm/"):
            name_ = k[9:]
            g = sd[name_].grad
            assert g is not None, name_
            assert abs(float(g.double().norm()) - float(v)) <= 1e-4 * max(1.0, float(v)), name_
            checked += 1
        if k.startswith("grad/"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), v, atol=2e-5, rtol=2e-4)
    assert checked > 100


def test_oracle_units():
    fix = load("units")
    for h in (1, 2, 4, 6, 8, 12):
        np.testing.assert_allclose(np.array(ref_cpu.alibi_slopes(h), dtype=np.float32), fix[f"alibi/slopes/{h}"], rtol=1e-6)
    np.testing.assert_array_equal(ref_cpu.alibi_bias(5, 5).numpy().astype(np.int32), fix["alibi/bias_5_5"])
    np.testing.assert_array_equal(ref_cpu.alibi_bias(1, 7).numpy().astype(np.int32), fix["alibi/bias_1_7"])
    z, y = torch.from_numpy(fix["mmd/z"]), torch.from_numpy(fix["mmd/y"])
    assert abs(float(ref_cpu.compute_mmd(z, y)) - float(fix["mmd/out"])) < 1e-6


def test_oracle_greedy_matches_reference_cached_decode():
    """The reference's KV/hidden-cached `unmask_tokens` (wrappers.py:325-407) equals the oracle's cache-free greedy
    loop token for token."""
    fix = load("tiny_greedy")
    cfg = model_config(preset="tiny", num_tokens=SMALL_VOCAB)
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB))
    sd = filled_state_dict(model, seed=3)
    tokens = torch.from_numpy(fix["in/tokens"])
    out = ref_cpu.greedy_unmask(sd, cfg, tokens, torch.from_numpy(fix["in/masked_perf"]),
                                torch.from_numpy(fix["out/score_embeddings"]), torch.from_numpy(fix["out/perf_embeddings"]))
    np.testing.assert_array_equal(out.numpy(), fix["out/tokens"])


def test_oracle_greedy_matches_reference_cached_decode_with_cross_attention():
    """Decoder layer blocks ('a','c','f') (context_emb_mode='attention'): the reference's cached `unmask_tokens` over a padded score
    (tests/golden/tiny_greedy_xattn.npz, oracle/refimport/make_golden_greedy_xattn.py) equals the oracle's greedy loop token for token
    -- INCLUDING the reference's defect in that mode (it reads a stale hidden row from the third decoded note on: see
    ref_cpu.greedy_unmask); with the intended row the two agree exactly on the notes decoded before the defect bites."""
    fix = load("tiny_greedy_xattn")
    kw = dict(preset="tiny", context_emb_mode="attention", num_tokens=SMALL_VOCAB)
    cfg = model_config(**kw)
    sd = filled_state_dict(ScorePerformer.init(model_config(**kw)), seed=4)
    out = ref_cpu.greedy_unmask(sd, cfg, torch.from_numpy(fix["in/tokens"]), torch.from_numpy(fix["in/masked_perf"]),
                                torch.from_numpy(fix["out/score_embeddings"]), torch.from_numpy(fix["out/perf_embeddings"]),
                                context_mask=torch.from_numpy(fix["in/score_mask"]), reference_hidden_row_defect=True)
    np.testing.assert_array_equal(out.numpy(), fix["out/tokens"])
    intended = ref_cpu.greedy_unmask(sd, cfg, torch.from_numpy(fix["in/tokens"]), torch.from_numpy(fix["in/masked_perf"]),
                                     torch.from_numpy(fix["out/score_embeddings"]), torch.from_numpy(fix["out/perf_embeddings"]),
                                     context_mask=torch.from_numpy(fix["in/score_mask"]))
    np.testing.assert_array_equal(intended.numpy()[:, :3], fix["out/tokens"][:, :3])
    assert (intended.numpy() != fix["out/tokens"]).any()


@pytest.mark.parametrize("tag", ["incl", "excl"])
def test_oracle_latent_dropout_matches_reference(tag):
    """Latent dropout as base.yaml:119-126 trains it (`latent_dropout=[0, .1, .2, .4]`): the reference's own per-level drop masks
    (`dropout_latent_mask`, mmd_transformer.py:537-542, recorded by oracle/refimport/make_golden_latent_dropout.py) fed to the oracle
    must reproduce the reference's embeddings after dropping, its inclusive / exclusive combined mask (mmd_transformer.py:249-253), the
    dead-pan exemption (mmd_transformer.py:284-291), the loss and every gradient norm -- the gradient has to pass through
    `embeddings * ~drop_mask` into the decoder's adaptive LayerNorms."""
    fix = load("latent_dropout")
    kw = dict(preset="tiny", num_tokens=SMALL_VOCAB, latent_dropout=[0.0, 0.1, 0.2, 0.4])
    cfg = model_config(**kw)
    cfg.perf_encoder.inclusive_latent_dropout = tag == "incl"
    sd = filled_state_dict(ScorePerformer.init(model_config(**kw)), seed=0)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v)
          for k, v in sd.items()}
    from oracle.weights import canonical
    leaves = {}
    for k in list(sd):
        c = canonical(k)
        sd[k] = leaves.setdefault(c, sd[k])
    batch = {k[3:]: torch.from_numpy(v) for k, v in fix.items() if k.startswith("in/")}
    assert bool(batch["deadpan_mask"].any()) and not bool(batch["perf_mask"].all())      # one dead-pan, ragged lengths
    z = [torch.from_numpy(fix[f"{tag}/z/{i}"]) for i in range(4)]
    drops = [torch.from_numpy(fix[f"{tag}/drop/{i}"]) if f"{tag}/drop/{i}" in fix else None for i in range(4)]
    assert drops[0] is None and all(d is not None and bool(d.any()) for d in drops[1:])
    out = ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True, drop_masks=drops)
    np.testing.assert_array_equal(out["perf_dropout_mask"].numpy(), fix[f"{tag}/dropout_mask"])
    np.testing.assert_allclose(out["perf_full_embeddings"].detach().numpy(), fix[f"{tag}/full_embeddings"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(out["perf_embeddings"].detach().numpy(), fix[f"{tag}/embeddings"], atol=3e-5, rtol=1e-4)
    assert (fix[f"{tag}/embeddings"] != fix[f"{tag}/full_embeddings"]).any()
    np.testing.assert_allclose(out["hidden_state"].detach().numpy(), fix[f"{tag}/hidden_state"], atol=3e-5, rtol=1e-4)
    assert abs(float(out["loss"]) - float(fix[f"{tag}/loss"])) < 2e-5
    pre = f"{tag}/losses/"
    ref_losses = {k[len(pre):]: float(v) for k, v in fix.items() if k.startswith(pre)}
    assert set(out["losses"]) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(float(out["losses"][k]) - v) < 2e-5, k
    out["loss"].backward()
    checked = 0
    for k, v in fix.items():
        if k.startswith(f"{tag}/gradnorm/"):
            name_ = k[len(tag) + 10:]
            g = sd[name_].grad
            assert g is not None, name_
            assert abs(float(g.double().norm()) - float(v)) <= 1e-4 * max(1.0, float(v)), name_
            checked += 1
        if k.startswith(f"{tag}/grad/"):
            np.testing.assert_allclose(sd[k[len(tag) + 6:]].grad.numpy(), v, atol=2e-5, rtol=2e-4)
    assert checked > 100
    if tag == "excl":      # the exclusive mask differs from the inclusive one on the same draws
        assert (fix["incl/dropout_mask"] != fix["excl/dropout_mask"]).any()
