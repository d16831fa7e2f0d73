"""Measure the per-step evaluator (N3) on the C3 train step: step time without metrics, with the evaluator fused into the
cross-entropy kernel (attach()), and with the unfused evaluator (second pass over the logits, like the reference's
trainer.py:462-464), same batch and weights.

    python tools/bench_eval.py [--batch 64] [--seq 2048] [--steps 8] [--out gpurun_out/eval.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scoreperformer_amd.arena import ParamArena, FusedAdamW          # noqa: E402
from scoreperformer_amd.models import ScorePerformer, ScorePerformerEvaluator   # noqa: E402
from scoreperformer_amd.synthetic import model_config, synthetic_batch          # noqa: E402

IGNORE = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(1234)
    cfg = model_config("c3", max_seq_len=a.seq, dropout=0.1)
    model = ScorePerformer.init(cfg)
    arena = ParamArena(model, dev)
    model.train()
    model.sync_free = True
    opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-6, grad_clip=2.0)
    batch = synthetic_batch(a.batch, a.seq, seed=1234, device=dev)
    model.perf_encoder.segment_bounds = {m: int(batch[k].max()) + 1 for m, k in
                                         (("bar_mean", "bars"), ("beat_mean", "beats"), ("onset_mean", "onsets"))}
    tv = {k: torch.linspace(0, 1, v).tolist() for k, v in cfg["num_tokens"].items()}
    ev = ScorePerformerEvaluator(model, ignore_keys=IGNORE, weighted_distance=True, token_values=tv)

    def run(mode):
        ev.attach() if mode == "fused" else ev.detach()
        last = None

        def step():
            out = model(**batch)
            out.loss.backward()
            opt.step()
            return ev(batch, out) if mode != "none" else None

        for _ in range(3):
            last = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            last = step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, ({k: float(v) for k, v in last.items()} if last else None)

    ms_none, _ = run("none")
    ms_fused, m_fused = run("fused")
    ms_plain, m_plain = run("unfused")
    res = {"workload": f"C3 train step b={a.batch} n={a.seq} + ScorePerformerEvaluator(weighted_distance=True, 4 predicted keys)",
           "ms_step_no_metrics": ms_none, "ms_step_fused_evaluator": ms_fused, "ms_step_unfused_evaluator": ms_plain,
           "evaluator_cost_ms": {"fused": ms_fused - ms_none, "unfused": ms_plain - ms_none},
           "metrics_fused": m_fused, "metrics_unfused": m_plain}
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
