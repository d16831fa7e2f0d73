"""Per-segment s_memtime sums of one attention-forward block (all 4 waves), from a build of attention.hip instrumented with TS() stamps
(tools/_bin/alt_attn_timing.so): where the cycles of a key-tile iteration go."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops, lib
b, h, n = 64, 8, 2048
dev = torch.device("cuda")
qkv = torch.randn(b, n, (h + 2) * 64, device=dev).bfloat16()
q = qkv[..., :h * 64].unflatten(-1, (h, 64)); k = qkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)); v = qkv[..., (h + 1) * 64:].unflatten(-1, (1, 64))
slopes = torch.full((h,), 2.0 ** -9, device=dev)
dbg = torch.zeros(64, device=dev, dtype=torch.int64)
lib.load().spn_attn_set_debug(ctypes.c_void_p(dbg.data_ptr()))
for causal in (False,):
    ops.attn_fwd(q, k, v, slopes=slopes, causal=causal)
    torch.cuda.synchronize()
    d = dbg.view(8, 8)[:4].cpu()
    names = ["loop top", "sync1 wait", "LDS store", "sync2 wait", "issue loads+classify", "S = K Q^T (reads + 16 MFMA)", "softmax", "PV (cvt + tr reads + 16 MFMA)"]
    iters = 32
    for w in range(4):
        tot = int(d[w].sum())
        print(f"wave {w}: total {tot} cycles = {tot / iters:.0f} per iteration; " + ", ".join(f"{names[i]} {int(d[w][i]) / iters:.0f}" for i in range(8)))
