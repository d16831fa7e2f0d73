"""Micro-benchmark of the bf16 GEMM at the C3 shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops

def run(M, N, K, ta, tb, f32, reps=10):
    dev = torch.device("cuda")
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    ops.gemm(a, b, ta=ta, tb=tb, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.gemm(a, b, ta=ta, tb=tb, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    am, bm = (a.t() if ta else a), (b if tb else b.t())
    o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    torch.matmul(am, bm, out=o2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.matmul(am, bm, out=o2)
    torch.cuda.synchronize()
    dl = (time.perf_counter() - t0) / reps
    print(f"M={M:6d} N={N:5d} K={K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'} {'f32' if f32 else 'bf16'}: {dt*1e3:7.3f} ms  {2.0*M*N*K/dt/1e12:6.0f} TF/s   | hipBLASLt (bf16 out) {dl*1e3:7.3f} ms {2.0*M*N*K/dl/1e12:6.0f} TF/s")

T = 131072
for shape in [(T, 4096, 512, False, False, False), (T, 512, 2048, False, False, True), (T, 640, 512, False, False, False),
              (T, 512, 512, False, False, True), (T, 2048, 512, False, True, False), (T, 512, 4096, False, True, False),
              (T, 512, 640, False, True, False), (4096, 512, T, True, True, True), (512, 2048, T, True, True, True),
              (640, 512, T, True, True, True), (8192, 8192, 8192, False, False, False)]:
    run(*shape)
