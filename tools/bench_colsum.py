import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda")
for N in (512, 640, 2048, 4096):
    x = torch.randn(131072, N, device=dev).bfloat16()
    for _ in range(3): ops.colsum(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.colsum(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    ref = x.float().sum(0)
    err = (ops.colsum(x) - ref).abs().max().item() / ref.abs().max().item()
    print(f"colsum N={N}: {t:.1f} us  {131072*N*2/t/1e6:.2f} TB/s  relerr {err:.1e}")
