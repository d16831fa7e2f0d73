"""GPU: the evaluator fused into the cross-entropy kernel (spn_ce_fwd_eval) against the reference evaluator's golden metrics, and
the attached evaluator on a model step against the CPU oracle."""
import numpy as np
import pytest
import torch

from test_evaluator_cpu import CASES

pytestmark = pytest.mark.gpu
TOL = 2e-5      # fp32 sums over <= ~100 rows here; the reference itself accumulates in fp32


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_sums_reproduce_the_reference_metrics(name):
    from scoreperformer_amd import ops
    c = CASES[name]
    cfg = c["cfg"]
    dev = torch.device("cuda")
    labels = torch.from_numpy(c["labels"]).to(dev)
    if cfg["mode"] in ("clm", "mixlm"):
        labels = labels[:, 1:]
    keys = list(c["logits"])
    tot = {"all": [0.0, 0.0], "pred": [0.0, 0.0]}
    for i, k in enumerate(keys):
        lg = torch.from_numpy(c["logits"][k]).to(dev)
        V = lg.shape[-1]
        tv = torch.from_numpy(c["values"][k]).to(dev) if cfg["with_values"] else None
        lse, sums, am, met = ops.ce_fwd(lg.reshape(-1, V), V, labels[..., i], want_argmax=True, eval_spec=(tv, cfg["weighted"]))
        assert torch.equal(am.view(lg.shape[:-1]).long(), lg.argmax(-1))
        valid, correct, dist = float(sums[1]), float(met[0]), float(met[1])
        assert valid == float((labels[..., i] != -100).sum())
        tot["all"][0] += correct; tot["all"][1] += valid
        if not (cfg["ignore_keys"] and k in cfg["ignore_keys"]):
            tot["pred"][0] += correct; tot["pred"][1] += valid
        if f"accuracy/{k}" in c["metrics"]:
            assert abs(correct / valid - c["metrics"][f"accuracy/{k}"]) < TOL
        else:
            assert valid == 0 or (cfg["ignore_keys"] and k in cfg["ignore_keys"])
        if f"distance/{k}" in c["metrics"]:
            want = c["metrics"][f"distance/{k}"]
            assert abs(dist / valid - want) < TOL * max(1.0, abs(want)), (k, dist / valid, want)
    assert abs(tot["all"][0] / tot["all"][1] - c["metrics"]["accuracy"]) < TOL
    if "accuracy/pred" in c["metrics"]:
        assert abs(tot["pred"][0] / tot["pred"][1] - c["metrics"]["accuracy/pred"]) < TOL


def test_fused_sums_on_bf16_logits_and_strided_labels():
    """bf16 logits with a padded row stride and labels taken as a strided column of [b, t, K] (how the LM head calls it)."""
    from oracle.evaluator_cpu import evaluate
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    b, t, V = 4, 300, 133
    buf = (torch.randn(b * t, 136, generator=g) * 4).to(torch.bfloat16).to(dev)
    lg = buf[:, :V]
    labels = torch.randint(0, V, (b, t, 5), generator=g)
    labels[torch.rand(b, t, generator=g) < 0.3] = -100
    tv = torch.rand(V, generator=g).sort().values
    lab = labels.to(dev)[..., 2]
    for weighted in (False, True):
        _, sums, _, met = ops.ce_fwd(lg, V, lab, eval_spec=(tv.to(dev), weighted))
        want = evaluate({"k": lg.float().cpu().numpy().reshape(b, t, V)}, labels[..., 2:3].numpy(), shift=False,
                        token_values={"k": tv.numpy()}, weighted_distance=weighted)
        assert abs(float(met[0] / sums[1]) - want["accuracy/k"]) < TOL
        assert abs(float(met[1] / sums[1]) - want["distance/k"]) < 5e-5


@pytest.mark.parametrize("weighted", [False, True])
def test_attached_evaluator_matches_oracle_on_a_model_step(weighted):
    from oracle.evaluator_cpu import evaluate
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer, ScorePerformerEvaluator
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    dev = torch.device("cuda")
    cfg = model_config("tiny", dropout=0.0)
    model = ScorePerformer.init(cfg)
    model.load_state_dict(filled_state_dict(model, seed=2))
    arena = ParamArena(model, dev)   # noqa: F841
    model.train()
    batch = synthetic_batch(3, 56, seed=4, ragged=True, device=dev)
    ignore = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]
    tv = {k: (torch.linspace(0, 1, v) ** 2 * 7).tolist() for k, v in cfg["num_tokens"].items()}
    ev = ScorePerformerEvaluator(model, ignore_keys=ignore, weighted_distance=weighted, token_values=tv).attach()
    torch.manual_seed(1)
    out = model(**batch)
    assert out.perf_decoder.eval_sums, "the attached evaluator must make the CE kernel produce the sums"
    fused = {k: float(v) for k, v in ev(batch, out).items()}
    out.loss.backward()                                     # the train step still differentiates
    # oracle on the very logits of this step
    logits = {k: v.float().cpu().numpy() for k, v in out.perf_decoder.logits.items()}
    want = evaluate(logits, batch["labels"].cpu().numpy(), shift=True, ignore_keys=ignore,
                    token_values={k: np.asarray(v, np.float32) for k, v in tv.items()}, weighted_distance=weighted)
    assert sorted(fused) == sorted(want)
    for k, w in want.items():
        assert abs(fused[k] - w) < 5e-5 * max(1.0, abs(w)), (k, fused[k], w)
    # and the unfused product path (no attach) reports the same
    ev.detach()
    torch.manual_seed(1)
    out2 = model(**batch)
    assert not out2.perf_decoder.eval_sums
    plain = {k: float(v) for k, v in ev(batch, out2).items()}
    assert sorted(plain) == sorted(fused)
    for k in plain:
        assert abs(plain[k] - fused[k]) < 5e-5 * max(1.0, abs(plain[k])), k
