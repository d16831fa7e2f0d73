"""Fused GLU input projection (spn_gemm_glu) against GEMM + activation kernel at the benchmark's FFN shape; prints one JSON line."""
import json
import sys

import torch

sys.path.insert(0, ".")
from scoreperformer_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    M, I, K = 131072, 2048, 512
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(2 * I, device="cuda") * 0.1
    out = {}
    timed(lambda: ops.gemm(x, w, tb=False, out_dtype=torch.bfloat16, bias=b), n=50)   # clocks up
    for p in (0.0, 0.1):
        fused = timed(lambda: ops.gemm_glu(x, w, b, act=0, p_drop=p, seed=5))
        gemm = timed(lambda: ops.gemm(x, w, tb=False, out_dtype=torch.bfloat16, bias=b))
        u = ops.gemm(x, w, tb=False, out_dtype=torch.bfloat16, bias=b)
        act = timed(lambda: ops.act_fwd(u, act=0, glu=True, p_drop=p, seed=5))
        out[f"p{p}"] = {"fused_us": round(fused, 1), "gemm_us": round(gemm, 1), "act_us": round(act, 1),
                        "fused_tflops": round(2 * M * 2 * I * K / fused / 1e6, 1)}
    # backward: input gradient of the output projection + activation backward (spn_gemm_glu_bwd) against the two kernels it replaces
    dy = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w2 = (torch.randn(K, I, device="cuda") * K ** -0.5).bfloat16()
    uu = torch.randn(M, 2 * I, device="cuda").bfloat16()
    cs = torch.zeros(2 * I, device="cuda")
    for p in (0.0, 0.1):
        fused = timed(lambda: ops.gemm_glu_bwd(dy, w2, uu, act=0, p_drop=p, seed=5, colsum=cs))
        gemm = timed(lambda: ops.gemm(dy, w2, tb=True, out_dtype=torch.bfloat16))
        dg = ops.gemm(dy, w2, tb=True, out_dtype=torch.bfloat16)
        act = timed(lambda: ops.act_bwd(uu, dg, act=0, glu=True, p_drop=p, seed=5, colsum=cs))
        out[f"bwd_p{p}"] = {"fused_us": round(fused, 1), "gemm_us": round(gemm, 1), "act_bwd_us": round(act, 1),
                            "fused_tflops": round(2 * M * I * K / fused / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
