"""Time the step's main GEMM shapes with the library named by SPN_LIB (A/B of kernel variants).  Prints TF/s per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
SHAPES = [  # M, N, K, ta, tb, f32out
    (131072, 4096, 512, 0, 0, 0), (4096, 512, 131072, 1, 1, 1), (131072, 512, 4096, 0, 1, 0), (131072, 2048, 512, 0, 1, 0),
    (131072, 512, 2048, 0, 0, 1), (131072, 640, 512, 0, 0, 0), (131072, 512, 512, 0, 0, 1), (8192, 8192, 8192, 0, 0, 0),
]
dev = torch.device("cuda")
res = []
for M, N, K, ta, tb, f32 in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res.append(f"{M}x{N}x{K}:{'T' if ta else 'N'}{'T' if tb else 'N'}:{'f32' if f32 else 'bf16'} {2*M*N*K/ms/1e9:7.1f}")
print(os.environ.get("SPN_LIB", "default"), " | ".join(res))
