"""Tuning aid: which call sites launch cast kernels in one train step (shape, dtypes, row mask, caller chain)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops, functional as F_
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
dev = torch.device("cuda:0")
cfg = model_config("c3", max_seq_len=2048, dropout=0.1, latent_dropout=[0, .1, .2, .4])
model = ScorePerformer.init(cfg); arena = ParamArena(model, dev); model.train(); model.sync_free = True
batch = synthetic_batch(8, 2048, seed=1, device=dev)
model.perf_encoder.segment_bounds = {m: int(batch[k].max()) + 1 for m, k in (("bar_mean", "bars"), ("beat_mean", "beats"), ("onset_mean", "onsets"))}
sites = collections.Counter()
orig = ops.cast
def cast(x, dtype, **kw):
    st = traceback.extract_stack(limit=8)
    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(st[:-1]) if "scoreperformer_amd" in f.filename)[:200]
    sites[(tuple(x.shape), str(x.dtype).replace("torch.", ""), str(dtype).replace("torch.", ""), kw.get("rowmask") is not None, where)] += 1
    return orig(x, dtype, **kw)
ops.cast = cast
F_.ops.cast = cast
for _ in range(2):
    sites.clear()
    out = model(**batch); out.loss.backward()
torch.cuda.synchronize()
for k, v in sorted(sites.items(), key=lambda kv: -kv[1] * kv[0][0][0] if kv[0][0] else 0):
    print(v, k)
