"""Exact-fp32 GEMM (spn_gemm_f32, both operands contiguous along K): the fp32-MFMA tile kernel against the VALU tile kernel on the shapes of
the render window's batched re-priming (decode.py RenderSession.prefill) -- python tools/bench_gemm_f32.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from scoreperformer_amd import lib, ops

dev = torch.device("cuda")
for M, N, K in [(512, 640, 512), (512, 512, 512), (512, 4096, 512), (512, 512, 2048), (280, 4096, 512), (128, 4096, 512), (2048, 4096, 512)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
    out = torch.empty(M, N, device=dev)
    res = []
    for knob in (0, 1):
        lib.set_tuning("gemm_f32_mfma", knob)
        for _ in range(3): ops.gemm_f32(a, w, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ops.gemm_f32(a, w, out=out)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e6)
    fl = 2.0 * M * N * K
    print(f"{M:5d} x {N:5d} x {K:5d}: VALU tiles {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF/s)   fp32 MFMA {res[1]:7.1f} us ({fl / res[1] / 1e6:6.1f} TF/s)")
