"""Golden losses of the REFERENCE for the third variant set (oracle/variants.py NAMES3; authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_variants3

One seeded training-mode forward + backward per variant, weights from oracle.weights (seed 1), batch from
synthetic_batch(2, 40, seed=5, ragged=True): stores the MMD prior draws, the loss, every loss-dict entry and the gradient norm of
every parameter (tests/golden/variants3.npz).  Data only.
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
from oracle.refimport import stubs  # noqa: E402

stubs.install()
warnings.filterwarnings("ignore")
from scoreperformer.models import Performer, ScorePerformer  # noqa: E402  (the reference)

from oracle.refimport.make_golden import RandnRecorder  # noqa: E402
from oracle.variants import NAMES3, SMALL_VOCAB, variant3_config  # noqa: E402
from oracle.weights import filled_state_dict  # noqa: E402
from scoreperformer_amd.synthetic import synthetic_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")


def main():
    out = {}
    batch = synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)
    for name in NAMES3:
        cls = Performer if name.startswith("performer_") else ScorePerformer
        model = cls.init(copy.deepcopy(variant3_config(name)))
        model.load_state_dict(filled_state_dict(model, seed=1), strict=True)
        model.train()
        with RandnRecorder() as rec:
            torch.manual_seed(3)
            res = model(**batch)
        res.loss.backward()
        out[f"{name}/loss"] = np.float64(float(res.loss))
        for k, v in res.losses.items():
            out[f"{name}/losses/{k}"] = np.float64(float(v))
        for i, z in enumerate(rec.samples):
            out[f"{name}/z{i}"] = z.numpy()
        seen = set()
        for k, p in model.named_parameters():
            if id(p) in seen or p.grad is None:
                continue
            seen.add(id(p))
            out[f"{name}/gradnorm/{k}"] = np.float64(float(p.grad.norm()))
        print(name, float(res.loss), {k: round(float(v), 4) for k, v in res.losses.items()}, "z draws", len(rec.samples),
              "grads", len(seen))
    path = os.path.join(OUT, "variants3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
