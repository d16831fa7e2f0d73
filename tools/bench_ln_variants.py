"""Which part of the LayerNorm backward separates it from tools/ln_bw_probe.hip (5.85 TB/s on the same access mix)?  Kernel-only timings
of the affine backward at T = 131072, D = 512 with pieces switched off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda"); T, D = 131072, 512
x = torch.randn(T, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
dy = torch.randn(T, D, device=dev).bfloat16(); dres = torch.randn(T, D, device=dev)
_, mean, rstd = ops.layernorm_fwd(x, g, b)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = {
    "affine + fork + dx16 + dgamma (the step's launch)": (dict(dres=dres, want_dx16=True, dgamma=dg, dbeta=db), 16),
    "same without dgamma / dbeta": (dict(dres=dres, want_dx16=True), 16),
    "same without the bf16 copy": (dict(dres=dres, dgamma=dg, dbeta=db), 14),
    "same without the residual gradient": (dict(want_dx16=True, dgamma=dg, dbeta=db), 12),
    "dx only": (dict(), 10),
}
for name, (kw, bpe) in cases.items():
    t = timeit(lambda: ops.layernorm_bwd(x, dy, g, None, mean, rstd, **kw))
    print(f"{name:55s} {t:7.1f} us  {T * D * bpe / t / 1e6:5.2f} TB/s")
