"""List every GEMM shape of one C3 train step with its measured time (HIP events around each launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
dev = torch.device("cuda")
b, n = int(os.environ.get("B", 64)), int(os.environ.get("N", 2048))
model = ScorePerformer.init(model_config("c3", dropout=float(os.environ.get("DROPOUT", 0.1)), latent_dropout=[0.0, 0.1, 0.2, 0.4])); arena = ParamArena(model, dev); model.train(); model.sync_free = True
batch = synthetic_batch(b, n, seed=1, device=dev)
model.perf_encoder.segment_bounds = {m: int(batch[k].max()) + 1 for m, k in (("bar_mean", "bars"), ("beat_mean", "beats"), ("onset_mean", "onsets"))}
opt = FusedAdamW(arena)
def step():
    out = model(**batch); out.loss.backward(); opt.step()
step(); torch.cuda.synchronize()
ops.PROFILE.enable(); step(); recs = ops.PROFILE.collect(); ops.PROFILE.disable()
agg = {}
for name, fl, ms, tag in recs:
    a = agg.setdefault((name, tag), [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
tot = sum(v[2] for v in agg.values())
for (name, tag), v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    print(f"{v[2]:8.2f} ms {100*v[2]/tot:5.1f}%  n={v[0]:3d}  {v[1]/(v[2]*1e-3)/1e12:7.1f} TF/s  {name} {tag}")
print("total", tot)
