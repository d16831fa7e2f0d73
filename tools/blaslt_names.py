"""Which kernels does the vendor library (hipBLASLt through torch.matmul) run on the step's plain GEMM shapes, and how fast?  Run under
rocprofv3 --kernel-trace --stats to see the Tensile kernel names (macro tile, wave tile, MFMA shape are encoded in them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda")
T = 131072
cases = [("dX ffn-in  NN", (T, 4096), (4096, 512), False, False),
         ("fwd qkv    NT", (T, 512), (640, 512), False, True),
         ("8192^3     NT", (8192, 8192), (8192, 8192), False, True),
         ("dW1        TN", (T, 4096), (T, 512), True, False),
         ("dW2        TN", (T, 512), (T, 2048), True, False)]
for name, sa, sb, ta, tb in cases:
    a = torch.randn(sa, device=dev).bfloat16(); b = torch.randn(sb, device=dev).bfloat16()
    A = a.t() if ta else a
    B = b.t() if tb else b
    for _ in range(3):
        c = torch.mm(A, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = torch.mm(A, B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    M, K = A.shape; N = B.shape[1]
    # ours: same math
    if ta:   # dW: A stored [K][M] (M-contiguous), B stored [K][N]
        o = lambda: ops.gemm(a, b, ta=True, tb=True, out_dtype=torch.float32)
    elif tb:
        o = lambda: ops.gemm(a, b, out_dtype=torch.bfloat16)
    else:
        o = lambda: ops.gemm(a, b, tb=True, out_dtype=torch.bfloat16)
    for _ in range(3):
        o()
    e0.record()
    for _ in range(10):
        o()
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 10
    print(f"{name}: M={M} N={N} K={K}: hipBLASLt (bf16 out) {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF/s | libspn {ms2 * 1e3:8.1f} us {2.0 * M * N * K / ms2 / 1e9:6.0f} TF/s", flush=True)
