"""What the zero-edit binding costs: the C3 train step driven exactly as the reference's trainer drives it --
model.to(device), torch.optim.AdamW(model.parameters()), GradScaler(enabled=False), clip_grad_norm_, scaler.step, zero_grad -- beside the
fast binding (ParamArena + FusedAdamW) on the same box, same batch, same dropout.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena, FusedAdamW
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch

dev = torch.device("cuda")
B, N, STEPS, WARM = int(os.environ.get("B", 64)), int(os.environ.get("N", 2048)), 8, 3
batch = synthetic_batch(B, N, seed=1234, device=dev, with_bounds=True)


def timed(step):
    for _ in range(WARM):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


def make():
    torch.manual_seed(1234)
    m = ScorePerformer.init(model_config("c3", max_seq_len=max(N, 256), dropout=0.1, latent_dropout=[0.0, 0.1, 0.2, 0.4]))
    m.train(); m.sync_free = True
    return m


res = {"workload": f"C3 train step, batch {B} x {N}, dropout 0.1"}
m = make(); m.to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=1e-6)
scaler = torch.cuda.amp.GradScaler(enabled=False)
params = list(m.parameters())


def ref_step():
    out = m(**batch)
    scaler.scale(out.loss).backward()
    scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(params, 2.0)
    scaler.step(opt); scaler.update(); opt.zero_grad()


res["zero_edit_ms_per_step"] = timed(ref_step)
del m, opt, params
torch.cuda.empty_cache()
m = make(); arena = ParamArena(m, dev)
fopt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-6, grad_clip=2.0)


def fast_step():
    m(**batch).loss.backward()
    fopt.step()


res["fast_binding_ms_per_step"] = timed(fast_step)
res["ratio"] = res["zero_edit_ms_per_step"] / res["fast_binding_ms_per_step"]
print(json.dumps(res))
