import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda"); T, I = 131072, 2048
u = torch.randn(T, 2 * I, device=dev).bfloat16(); d = torch.randn(T, I, device=dev).bfloat16()
x32 = torch.randn(T, 512, device=dev)
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for p in (0.0, 0.1):
    t = timeit(lambda: ops.act_fwd(u, act=0, glu=True, p_drop=p, seed=3)); print(f"act_fwd p={p}: {t:.1f} us {T*I*6/t/1e6:.2f} TB/s")
    t = timeit(lambda: ops.act_bwd(u, d, act=0, glu=True, p_drop=p, seed=3)); print(f"act_bwd p={p}: {t:.1f} us {T*I*10/t/1e6:.2f} TB/s")
t = timeit(lambda: ops.cast(x32, torch.bfloat16)); print(f"cast f32->bf16 [T,512]: {t:.1f} us {T*512*6/t/1e6:.2f} TB/s")
