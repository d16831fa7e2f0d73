"""Time the embedding forward / backward kernels at the step's shape (T = 131072, 12 keys x 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
from scoreperformer_amd.synthetic import PERFORMANCE_VOCAB
dev = torch.device("cuda"); B, n = 64, 2048
V = list(PERFORMANCE_VOCAB.values())
tables = [torch.randn(v, 128, device=dev) for v in V]
tokens = torch.stack([torch.randint(0, v, (B, n), device=dev) for v in V], -1)
gamma, beta = torch.ones(1536, device=dev), torch.zeros(1536, device=dev)
def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = timeit(lambda: ops.embed_fwd(tables, tokens, gamma, beta))
print(f"embed_fwd: {t:.1f} us  ({B*n*1536*2/t/1e6:.2f} TB/s of bf16 output)")
y, mean, rstd = ops.embed_fwd(tables, tokens, gamma, beta)[:3]
dy = torch.randn(B * n, 1536, device=dev).bfloat16()
dts = [torch.zeros_like(t_) for t_ in tables]
t = timeit(lambda: ops.embed_bwd(tables, tokens, dy, gamma, mean, rstd, dgamma=torch.zeros(1536, device=dev), dbeta=torch.zeros(1536, device=dev)), reps=10)
print(f"embed_bwd (stats + scatter): {t:.1f} us")
