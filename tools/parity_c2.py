"""Prints the C2-scale parity table of tests/test_parity_c2_gpu.py: loss, every loss-dict entry, and the relative L2 gradient error of
EVERY parameter tensor against the fp32 CPU oracle (sorted, worst first)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_parity_c2_gpu import FLOOR, GRAD_BOUNDS, REL, all_grad_names, grad_errors, run_c2

dev = torch.device("cuda:0")
model, out, ref, sdg = run_c2(dev)
print(f"loss HIP {float(out.loss):.6f}  CPU {float(ref['loss']):.6f}  |diff| {abs(float(out.loss) - float(ref['loss'])):.2e}")
for k, v in ref["losses"].items():
    print(f"  {k:24s} HIP {float(out.losses[k]):.6f} CPU {float(v):.6f} diff {float(out.losses[k]) - float(v):+.2e}")
names = all_grad_names(model, sdg)
rows = sorted(grad_errors(model, sdg, names), key=lambda r: -r[1])
err2 = sum((e * n) ** 2 for _, e, n in rows) ** 0.5
ref2 = sum(n ** 2 for _, _, n in rows) ** 0.5
print(f"{len(rows)} tensors; relative L2 gradient error: max {rows[0][1]:.4f}, median {rows[len(rows) // 2][1]:.4f}; whole gradient "
      f"{err2 / ref2:.4f} (|g| = {ref2:.4f}); rule ||d|| <= {REL} ||g|| + {FLOOR}: worst margin "
      f"{max(e * n / (REL * n + FLOOR) for _, e, n in rows):.3f} of the bound")
for k, e, n in rows[:25]:
    print(f"  {e:.4f}  |g|={n:.3e}  {k}")
print("tensors named in the test:")
for k, e, n in grad_errors(model, sdg, list(GRAD_BOUNDS)):
    print(f"  {e:.4f}  bound {GRAD_BOUNDS[k]:.3f}  |g|={n:.3e}  {k}")
