"""In-flight sensitivity of the ping-pong GEMM (round 5): the same shapes timed with the library as shipped (4 half-tiles = 64 KiB of LDS
DMA in flight per workgroup at the steady-state wait) and with variants that wait for more (tools/build_variant.py gemm.hip vm6_spn.so
-DPP_STEADY_VM=6 / vm4: 3 / 2 half-tiles in flight).  If time rises steeply as the in-flight depth falls, the main loop is bound by
bytes in flight / latency and a deeper ring would pay; if it does not move, it is not.  One library per process: SPN_LIB selects it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops

dev = torch.device("cuda")
SHAPES = [(131072, 512, 4096, 0, 1, 0, "dX of the FFN input projection (long K, NT)"),
          (4096, 512, 131072, 1, 1, 1, "dW1 (split-K, TT, fp32)"),
          (512, 2048, 131072, 1, 1, 1, "dW2 (split-K, TT, fp32)"),
          (131072, 512, 2048, 0, 0, 1, "FFN output projection (NN, fp32 + residual-less)"),
          (131072, 4096, 512, 0, 0, 0, "K = 512 projection (NN, bf16)"),
          (8192, 8192, 8192, 0, 0, 0, "8192^3")]
print("library:", os.environ.get("SPN_LIB", "shipped"), " gemm_ow =", os.environ.get("SPN_GEMM_OW", "0"))
for M, N, K, ta, tb, f32, what in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"  {M:6d} x {N:5d} x {K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'} {'f32' if f32 else 'bf16'}: {best * 1e3:8.1f} us  {2.0 * M * N * K / best / 1e9:7.0f} TF/s   {what}")
    del a, b, out
