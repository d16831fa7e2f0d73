"""Time LayerNorm forward/backward at the step's shape (T = 131072, D = 512) -> effective TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda"); T, D = 131072, 512
x = torch.randn(T, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
gb = torch.randn(T, 2 * D, device=dev)
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
y = torch.empty(T, D, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: ops.layernorm_fwd(x, g, b, out=y))
print(f"ln_fwd affine  fp32->bf16: {t:7.1f} us  {T*D*6/t/1e6:5.2f} TB/s")
t = timeit(lambda: ops.layernorm_fwd(x, None, None, gb, out=y))
print(f"ln_fwd adaptive fp32->bf16: {t:7.1f} us  {T*D*(6+8)/t/1e6:5.2f} TB/s")
# backward at the step's shapes: affine with fork (d_residual), fp32 dx + bf16 copy
dy = torch.randn(T, D, device=dev).bfloat16(); dres = torch.randn(T, D, device=dev)
_, mean, rstd = ops.layernorm_fwd(x, g, b)
import inspect
def bwd_affine():
    return ops.layernorm_bwd(x, dy, g, None, mean, rstd, dres=dres, want_dx16=True, dgamma=torch.zeros(D, device=dev), dbeta=torch.zeros(D, device=dev))
try:
    t = timeit(bwd_affine); print(f"ln_bwd affine+fork: {t:7.1f} us  {T*D*(4+2+4+4+2)/t/1e6:5.2f} TB/s")
except TypeError as e:
    print("ln_bwd signature:", inspect.signature(ops.layernorm_bwd))
