"""Golden losses of the REFERENCE model for the ablation recipes' model variants (authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_ablations

One seeded training-mode forward per variant (oracle/variants.py), weights from oracle.weights (seed 1), batch from
synthetic_batch(2, 40, seed=5, ragged=True): stores the MMD prior draws, the loss and every loss-dict entry
(tests/golden/ablations.npz).  Data only.
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
from scoreperformer.models import ScorePerformer  # noqa: E402  (the reference)

from oracle.refimport.make_golden import RandnRecorder  # noqa: E402
from oracle.variants import NAMES, SMALL_VOCAB, ablation_config, variant_batch  # noqa: E402
from oracle.weights import filled_state_dict  # noqa: E402
from scoreperformer_amd.synthetic import synthetic_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")


def main():
    out = {}
    batch = synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)
    for name in NAMES:
        model = ScorePerformer.init(copy.deepcopy(ablation_config(name)))
        model.load_state_dict(filled_state_dict(model, seed=1), strict=True)
        model.train()
        with RandnRecorder() as rec:
            torch.manual_seed(3)
            res = model(**variant_batch(name, batch))
        out[f"{name}/loss"] = np.float64(float(res.loss))
        for k, v in res.losses.items():
            out[f"{name}/losses/{k}"] = np.float64(float(v))
        for i, z in enumerate(rec.samples):
            out[f"{name}/z{i}"] = z.numpy()
        print(name, float(res.loss), {k: round(float(v), 4) for k, v in res.losses.items()}, "z draws", len(rec.samples))
    path = os.path.join(OUT, "ablations.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
