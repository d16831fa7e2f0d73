"""Margins of tests/test_parity_c2_gpu.py on this box: loss difference, whole-gradient relative L2, the worst tensor against the per-tensor rule.
    python tools/parity_measure.py"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_parity_c2_gpu as T
dev = torch.device("cuda:0")
for preset, seq, seed in (("c2", 1024, 21), ("c3", 2048, 33)):
    model, out, ref, sdg = T.run_c2(dev, preset=preset, seq=seq, seed=seed)
    names = T.all_grad_names(model, sdg)
    rows = T.grad_errors(model, sdg, names)
    err2 = sum((e * n) ** 2 for _, e, n in rows) ** 0.5; ref2 = sum(n ** 2 for _, _, n in rows) ** 0.5
    worst = max((e * n) / (T.REL * n + T.FLOOR) for _, e, n in rows)
    import statistics
    print(preset, "loss diff", abs(float(out.loss) - float(ref["loss"])), "whole-vector rel", err2 / ref2, "worst per-tensor fraction of bound", worst,
          "median rel", statistics.median(e for _, e, _ in rows), "max rel (norm>1e-2)", max(e for _, e, n in rows if n > 1e-2))
