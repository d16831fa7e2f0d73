"""Run one GEMM shape repeatedly (for rocprofv3 counter passes).  env: M N K TA TB F32 REPS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
E = lambda k, d: int(os.environ.get(k, d))
M, N, K, ta, tb, f32, reps = E("M", 8192), E("N", 8192), E("K", 8192), E("TA", 0), E("TB", 0), E("F32", 0), E("REPS", 5)
dev = torch.device("cuda")
a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
if E("GLU", 0):   # N = gated outputs: b is [2N, K]
    b = torch.randn(2 * N, K, device=dev).bfloat16()
    for _ in range(reps):
        ops.gemm_glu(a, b, None, act=0, p_drop=0.1, seed=3)
else:
    for _ in range(reps):
        ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out)
torch.cuda.synchronize()
