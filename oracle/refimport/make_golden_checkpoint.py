"""A checkpoint file written by the REFERENCE (run in the authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_checkpoint

Builds the reference's ScorePerformer at a micro size, takes two AdamW steps with the reference's `Optimizer` wrapper's inner
torch.optim.AdamW on a seeded batch, and saves the dict of `Trainer._save_checkpoint` (trainer.py:296-314) with torch.save to
tests/golden/micro_checkpoint.pt, plus the loss of a third forward (tests/golden/micro_checkpoint_probe.npz) so a loader can
prove it restored a working model.  Data only (tensors, config dict, JSON strings).
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
from oracle.refimport import stubs  # noqa: E402

stubs.install()
warnings.filterwarnings("ignore")
from scoreperformer.models import ScorePerformer  # noqa: E402  (the reference)

from oracle.refimport.make_golden import SMALL_VOCAB, RandnRecorder  # noqa: E402
from oracle.weights import filled_state_dict  # noqa: E402
from scoreperformer_amd.synthetic import model_config, synthetic_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
MICRO = dict(preset="tiny", num_tokens=SMALL_VOCAB, dim=32, heads=2, depths=(1, 1, 1), emb_dims=8, latent_dim=[8, 4, 2, 2], max_seq_len=64)


def main():
    cfg = model_config(**MICRO)
    model = ScorePerformer.init(model_config(**MICRO))
    model.load_state_dict(filled_state_dict(model, seed=21), strict=True)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.99)
    batch = synthetic_batch(2, 24, seed=31, ragged=True, num_tokens=SMALL_VOCAB)
    for _ in range(2):
        torch.manual_seed(7)
        opt.zero_grad()
        model(**batch).loss.backward()
        opt.step()
    sched.step()
    plain = json.loads(json.dumps(cfg, default=lambda o: dict(o)))
    checkpoint = {
        "experiment": {"config": json.dumps({"model": plain}), "trainer": json.dumps({"output_dir": "results"}),
                       "state": json.dumps({"epoch": 1, "global_step": 2})},
        "model": {"config": plain, "state_dict": model.state_dict()},
        "optimizer": {"optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict()},
    }
    path = os.path.join(OUT, "micro_checkpoint.pt")
    torch.save(checkpoint, path)
    model.eval()
    with RandnRecorder() as rec, torch.no_grad():
        torch.manual_seed(9)
        out = model(**batch)
    np.savez_compressed(os.path.join(OUT, "micro_checkpoint_probe.npz"), loss=float(out.loss),
                        **{f"loss/{k}": float(v) for k, v in out.losses.items()}, **{f"z{i}": z.numpy() for i, z in enumerate(rec.samples)},
                        **{f"batch/{k}": v.numpy() for k, v in batch.items()})
    n = sum(p.numel() for p in model.parameters())
    print("wrote", path, os.path.getsize(path), "bytes;", n, "parameters; probe loss", float(out.loss), "z draws", len(rec.samples))


if __name__ == "__main__":
    main()
