"""Run-to-run noise of one tiny train step (loss entries, logits, gradients) with autocast off / off / on, both bindings: the tolerances of
tests/test_amp_gpu.py come from here (sums through float atomics differ by 1-3e-7 between two plain runs; logits are bit-identical).
    python tools/det_probe.py"""
import sys, torch
sys.path.insert(0, ".")
from oracle.weights import filled_state_dict
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
dev = torch.device("cuda:0")
for arena in (False, True):
    cfg = model_config("tiny", dropout=0.0)
    model = ScorePerformer.init(model_config("tiny", dropout=0.0))
    model.load_state_dict(filled_state_dict(model, seed=11))
    if arena: ar = ParamArena(model, dev)
    else: model.to(dev)
    model.train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(2, 64, seed=3, ragged=True).items()}
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)).to(dev) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    runs = []
    for mode in ("off", "off", "on"):
        if arena: ar.zero_grad()
        else: model.zero_grad(set_to_none=True)
        model.perf_encoder._z_override = z
        with torch.autocast("cuda", dtype=torch.float16, enabled=(mode == "on")):
            out = model(**batch)
        (out.loss * 65536.0).backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        runs.append((out.loss.detach().clone(), {k: v.detach().clone() for k, v in out.losses.items()},
                     {k: v.detach().clone() for k, v in out.perf_decoder.logits.items()}, grads))
    for name, (a, b) in (("off-vs-off", (runs[0], runs[1])), ("off-vs-on", (runs[0], runs[2]))):
        print(f"== arena={arena} {name}: loss equal {torch.equal(a[0], b[0])}")
        for k in a[1]:
            if not torch.equal(a[1][k], b[1][k]): print("   losses", k, float(a[1][k]), float(b[1][k]), float((a[1][k]-b[1][k]).abs()))
        for k in a[2]:
            if not torch.equal(a[2][k], b[2][k]): print("   logits", k, a[2][k].dtype, b[2][k].dtype, float((a[2][k].float()-b[2][k].float()).abs().max()))
        nd = [(k, float((a[3][k]-b[3][k]).norm()/a[3][k].norm().clamp_min(1e-30))) for k in a[3] if not torch.equal(a[3][k], b[3][k])]
        print("   grads differing:", len(nd), "of", len(a[3]), sorted(nd, key=lambda t: -t[1])[:5])
