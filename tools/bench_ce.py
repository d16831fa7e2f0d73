"""Time the cross-entropy forward (spn_ce_fwd) at the step's shape: T = 131008 rows, the vocabularies of the four predicted keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
dev = torch.device("cuda"); T = 131008
for V in (165, 85, 133):
    lg = torch.randn(T, 168, device=dev)[:, :V]
    lab = torch.randint(0, V, (64, 2047), device=dev); lab[torch.rand(64, 2047, device=dev) < 0.3] = -100
    f = lambda: ops.ce_fwd(lg, V, lab, want_argmax=True)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(V, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us")
