"""Race / edge screen of spn_gemm_bf16: random shapes (every layout, bf16 / fp32 output, bias, residual, split-K through a workspace),
each compared IN FULL with torch's fp32 matmul of the same bf16 operands.  Usage: python tools/stress_gemm.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops



def run(cases: int = 200, seed: int = 0, verbose: bool = True) -> int:
    rng = random.Random(seed)
    dev = torch.device("cuda")
    bad = 0
    for c in range(cases):
        ta, tb, f32 = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.5
        if rng.random() < 0.3:   # weight-gradient like: small M x N, long K (split-K)
            M, N, K = 8 * rng.randint(16, 160), 8 * rng.randint(16, 96), 64 * rng.randint(64, 512)
            ta = tb = True; f32 = True
        else:
            M, N, K = 8 * rng.randint(16, 700), 8 * rng.randint(16, 200), 64 * rng.randint(4, 40)
        g = torch.Generator(device=dev).manual_seed(c)
        a = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).bfloat16()
        b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
        bias = torch.randn(N, device=dev, generator=g) if rng.random() < 0.5 else None
        res = torch.randn(M, N, device=dev, generator=g) if (f32 and rng.random() < 0.5) else None
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        ops.gemm(a, b, ta=ta, tb=tb, out=out, bias=bias, residual=res)
        ref = (a.t() if ta else a).float() @ (b if tb else b.t()).float()
        if bias is not None:
            ref = ref + bias
        if res is not None:
            ref = ref + res
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        tol = 1e-2 if not f32 else 2e-5 * (K ** 0.5)
        ok = err < tol and bool(torch.isfinite(out.float()).all())
        if not ok:
            bad += 1
            print(f"MISMATCH case {c}: M={M} N={N} K={K} ta={ta} tb={tb} f32={f32} bias={bias is not None} res={res is not None} err={err:.3e}")
    if verbose:
      print(f"{cases} cases, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
