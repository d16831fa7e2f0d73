"""Sanity of the whole train step over a few hundred optimizer steps (C2 model, 16 x 1024 synthetic notes, dropout 0.1, latent dropout on):
the loss must fall from its initial value and stay finite.   python tools/train_sanity.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
torch.manual_seed(0)
model = ScorePerformer.init(model_config("c2", dropout=0.1, latent_dropout=[0.0, 0.1, 0.2, 0.4]))
arena = ParamArena(model, dev)
model.train()
model.sync_free = True
opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-6, grad_clip=2.0)
batches = [synthetic_batch(16, 1024, seed=100 + i, ragged=True, deadpan_p=0.25, device=dev, with_bounds=True) for i in range(8)]
hist = []
for s in range(steps):
    out = model(**batches[s % len(batches)])
    out.loss.backward()
    opt.step()
    if s % 25 == 0 or s == steps - 1:
        hist.append((s, float(out.loss), {k: round(float(v), 4) for k, v in out.losses.items() if k in ("Velocity", "MMD", "MMD/onset_mean", "MMD/onset_mean/deadpan")}))
        print(hist[-1], flush=True)
assert all(torch.isfinite(torch.tensor(h[1])) for h in hist)
assert hist[-1][1] < hist[0][1] - 0.2, (hist[0], hist[-1])
print("ok: loss", hist[0][1], "->", hist[-1][1])
