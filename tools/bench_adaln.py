"""Fused AdaptiveLayerNorm (projection inside the kernel) against the unfused path (K = 64 GEMM + LayerNorm with bf16 gamma | beta rows)
at the step's shape (T = 131008, D = 512, C = 64): correctness against fp32 torch and time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda"); T, D, C = int(os.environ.get("T", 131008)), 512, 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(T, D, device=dev, generator=g) * 2 + 0.3
cond = torch.randn(T, C, device=dev, generator=g).bfloat16()
w = (torch.randn(2 * D, C, device=dev, generator=g) * 0.1).bfloat16()
bias = torch.cat([torch.ones(D), torch.zeros(D)]).to(dev) + 0.05 * torch.randn(2 * D, device=dev, generator=g)
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def unfused():
    gb = ops.gemm(cond, w, out_dtype=torch.bfloat16, bias=bias)
    return ops.layernorm_fwd(x, None, None, gb)
y, mean, rstd, gam = ops.adaln_fwd(x, cond, w, bias)
gbr = cond.float() @ w.float().t() + bias
ref = gbr[:, :D] * torch.nn.functional.layer_norm(x, (D,)) + gbr[:, D:]
y0, m0, r0 = unfused()
err = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
print("fused  vs fp32 torch:", err(y, ref), " mean", err(mean, x.mean(1)), " rstd", err(rstd, (x.var(1, unbiased=False) + 1e-5).rsqrt()))
print("unfused vs fp32 torch:", err(y0, ref), " gamma rows vs fp32:", err(gam, gbr[:, :D]))
print(f"fused fwd {timeit(lambda: ops.adaln_fwd(x, cond, w, bias)):7.1f} us   unfused (gemm + ln) {timeit(unfused):7.1f} us")

