"""Golden vectors of the reference's ScorePerformerEvaluator (run in the authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_evaluator

Writes tests/golden/evaluator.npz: random logits / labels / token values (inputs) and the metrics the REAL reference evaluator
returns for them, for weighted and plain distances, with and without ignored keys.  Data only.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden")   # module-level, like the other generators: a checker may point it elsewhere
sys.path.insert(0, ROOT)
from oracle.refimport import stubs  # noqa: E402

stubs.install()
sys.path.insert(0, stubs.REFERENCE_ROOT)
from scoreperformer.models.scoreperformer.evaluator import ScorePerformerEvaluator  # noqa: E402
from scoreperformer.models.scoreperformer.wrappers import ScorePerformerLMModes      # noqa: E402

VOCAB = {"Bar": 20, "Position": 33, "Pitch": 92, "Velocity": 132, "Duration": 70, "Tempo": 125, "TimeSig": 9, "PositionShift": 17,
         "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 165, "RelPerfDuration": 85}
IGNORE = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]   # base.yaml evaluator
CASES = {
    # name: (mode, ignore_keys, weighted, with token values)
    "mixlm_weighted": ("mixlm", IGNORE, True, True),
    "mixlm_plain": ("mixlm", IGNORE, False, True),
    "mlm_all_keys": ("mlm", None, False, True),
    "clm_no_values": ("clm", IGNORE, False, False),
}


class FakeTokenizer:
    def __init__(self, values):
        self.values = values

    def token_values(self, normalize=False):
        assert normalize is False
        return self.values


def main():
    rng = np.random.default_rng(777)
    b, t = 3, 29
    out = {}
    for name, (mode, ignore, weighted, with_values) in CASES.items():
        logits = {k: (rng.standard_normal((b, t - 1 if mode != "mlm" else t, v)) * 3).astype(np.float32) for k, v in VOCAB.items()}
        labels = np.stack([rng.integers(0, v, size=(b, t)) for v in VOCAB.values()], -1).astype(np.int64)
        # make the targets partly predictable so accuracies are not ~0
        for i, k in enumerate(VOCAB):
            hit = rng.random((b, logits[k].shape[1])) < 0.4
            lab = labels[:, -logits[k].shape[1]:, i]
            bi, ti = np.nonzero(hit)
            logits[k][bi, ti, lab[bi, ti]] += 12.0
        labels[rng.random((b, t)) < 0.25] = -100                         # padded / special rows
        for i, k in enumerate(VOCAB):
            if k in IGNORE and name != "mlm_all_keys":
                labels[..., i] = -100                                   # label_pad_ignored_dims=True
        values = {k: np.sort(rng.random(v) * (100 if k == "Tempo" else 4)).astype(np.float32) for k, v in VOCAB.items()}
        model = SimpleNamespace(mode=ScorePerformerLMModes(mode))
        ev = ScorePerformerEvaluator(model, tokenizer=FakeTokenizer(values) if with_values else None, weighted_distance=weighted,
                                     ignore_keys=ignore)
        outputs = SimpleNamespace(logits={k: torch.from_numpy(v) for k, v in logits.items()}, hidden_state=torch.zeros(1))
        metrics = ev({"labels": torch.from_numpy(labels)}, outputs)
        out[f"{name}/cfg"] = np.array(repr(dict(mode=mode, ignore_keys=ignore, weighted=weighted, with_values=with_values)))
        out[f"{name}/labels"] = labels
        for k in VOCAB:
            out[f"{name}/logits/{k}"] = logits[k]
            out[f"{name}/values/{k}"] = values[k]
        for k, v in metrics.items():
            out[f"{name}/metric/{k}"] = np.float64(float(v))
        print(name, {k: round(float(v), 4) for k, v in metrics.items()})
    path = os.path.join(OUT, "evaluator.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
