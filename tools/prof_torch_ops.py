"""List the non-spn (ATen) GPU ops of one C3 train step by shape (where do stray adds/fills/copies come from)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
dev = torch.device("cuda")
b, n = int(os.environ.get("B", 64)), int(os.environ.get("N", 2048))
model = ScorePerformer.init(model_config("c3", dropout=0.1)); arena = ParamArena(model, dev); model.train(); model.sync_free = True
batch = synthetic_batch(b, n, seed=1, device=dev)
model.perf_encoder.segment_bounds = {m: int(batch[k].max()) + 1 for m, k in (("bar_mean", "bars"), ("beat_mean", "beats"), ("onset_mean", "onsets"))}
opt = FusedAdamW(arena)
def step():
    out = model(**batch); out.loss.backward(); opt.step()
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:90], [s for s in e.stack if "scoreperformer_amd" in s or "autograd" in s][:3]))
rows.sort(reverse=True)
for t, c, k, sh, st in rows[:40]:
    print(f"{t/1e3:8.2f} ms n={c:4d} {k:28s} {sh}")
    for s_ in st:
        print("        ", s_[-110:])
