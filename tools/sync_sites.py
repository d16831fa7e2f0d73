"""Where does one composed train step wait for the device?  torch.cuda.set_sync_debug_mode("warn") around forward + backward + step of a
collated tiny batch and of the bench's C3 step; every warning is printed with the python stack that raised it."""
import os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch

dev = torch.device("cuda")
preset = sys.argv[1] if len(sys.argv) > 1 else "tiny"
model = ScorePerformer.init(model_config(preset, dropout=0.1))
arena = ParamArena(model, dev); model.train(); model.sync_free = True
opt = FusedAdamW(arena, lr=1e-3, weight_decay=1e-2, grad_clip=2.0)
batch = synthetic_batch(2, 256 if preset == "tiny" else 1024, seed=3, ragged=True, device=dev, with_bounds=True)


def show(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", message, flush=True)
    traceback.print_stack(limit=14)


warnings.showwarning = show
warnings.simplefilter("always")
for i in range(3):
    if i:
        torch.cuda.set_sync_debug_mode("warn")
    out = model(**batch); out.loss.backward(); opt.step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("step", i, float(out.loss), flush=True)
