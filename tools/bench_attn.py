"""Micro-benchmark of the attention kernels at the C3 layer shape (b=64, h=8 MQA, n=2048, dh=64)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops

def main():
    b, h, n = int(os.environ.get("B", 64)), 8, int(os.environ.get("N", 2048))
    reps = int(os.environ.get("REPS", 5))
    dev = torch.device("cuda")
    qkv = torch.randn(b, n, (h + 2) * 64, device=dev).bfloat16()
    q = qkv[..., :h * 64].unflatten(-1, (h, 64)); k = qkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)); v = qkv[..., (h + 1) * 64:].unflatten(-1, (1, 64))
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(h)], device=dev)
    if os.environ.get("SLOPES") == "0":
        slopes = None
    elif os.environ.get("SLOPES") == "flat":   # no head reaches the band limit: isolates the diagonal-tile cost from the band skip
        slopes = torch.full((h,), 2.0 ** -9, device=dev)
    d_o = torch.randn(b, n, h, 64, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    dq = dqkv[..., :h * 64].unflatten(-1, (h, 64)); dk = dqkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)); dv = dqkv[..., (h + 1) * 64:].unflatten(-1, (1, 64))
    kmask = None
    if os.environ.get("RAGGED") == "1":   # SURVEY 8(d)'s ragged variant: lengths ~ U{n/2..n}, right-padded
        gen = torch.Generator().manual_seed(0)
        lens = torch.randint(n // 2, n + 1, (b,), generator=gen)
        kmask = (torch.arange(n)[None, :] < lens[:, None]).to(dev)
        print(f"ragged: valid fraction {kmask.float().mean().item():.3f}")
    if os.environ.get("LEN_FRAC"):   # every sequence the same length: separates the cost of padding from the imbalance of a ragged batch
        kmask = (torch.arange(n)[None, :] < int(n * float(os.environ["LEN_FRAC"]))).expand(b, n).contiguous().to(dev)
    if os.environ.get("FULLMASK") == "1":   # a mask tensor without a single False: what a full-length batch hands the kernels
        kmask = torch.ones(b, n, dtype=torch.bool, device=dev)
    qmask = kmask if os.environ.get("QMASK", "1") == "1" else None   # self-attention: the padding rows are the padding keys
    import inspect
    qkw = {"qmask": qmask} if "qmask" in inspect.signature(getattr(ops, "_attn_fwd_raw", ops.attn_fwd)).parameters else {}   # older trees (A/B)
    p_drop = float(os.environ.get("DROP", 0))   # attention dropout (the benchmark's 0.1)
    dkw = {"p_drop": p_drop, "seed": 7} if p_drop > 0 else {}
    for causal in (False, True):
        fl = 4.0 * b * h * n * n * 64 * (0.5 if causal else 1.0)
        res = ops.attn_fwd(q, k, v, slopes=slopes, causal=causal, kmask=kmask, **qkw, **dkw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = ops.attn_fwd(q, k, v, slopes=slopes, causal=causal, kmask=kmask, **qkw, **dkw)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / reps
        o, lse = res[0], res[1]
        bkw = {"p_drop": p_drop, "dropbits": res[2]} if p_drop > 0 else {}
        ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, **qkw, slopes=slopes, causal=causal, want_dslope=(slopes is not None and os.environ.get("DSLOPE", "1") != "0"), **bkw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, **qkw, slopes=slopes, causal=causal, want_dslope=(slopes is not None and os.environ.get("DSLOPE", "1") != "0"), **bkw)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / reps
        print(f"causal={causal}: fwd {tf*1e3:.3f} ms {fl/tf/1e12:.0f} TF/s | bwd {tb*1e3:.3f} ms {2.5*fl/tb/1e12:.0f} TF/s (5-matmul flops)")

main()
