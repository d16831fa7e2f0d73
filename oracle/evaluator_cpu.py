"""CPU restatement (numpy, float64 accumulation) of the reference's per-step metric evaluator -- TEST INFRASTRUCTURE, not product code.

Parity PINNED: `tests/golden/evaluator.npz` holds logits / labels / token values and the metrics the reference's own
`ScorePerformerEvaluator` returned for them (`oracle/refimport/make_golden_evaluator.py`, run with the real reference imported in
the authoring container); `tests/test_evaluator_cpu.py` checks this file against them.

Follows /root/reference/scoreperformer/models/scoreperformer/evaluator.py:
  :38-40   _accuracy (mean of prediction == label over labels != pad)
  :41-42   _distance (mean |tv[pred] - tv[label]|)
  :44-45   _weighted_distance (mean over rows of sum_c softmax_c |tv[label] - tv[c]|)
  :62-63   CLM / MixedLM modes drop the first label position
  :68-71   predictions = per-key argmax (first index on ties)
  :73-86   accuracy, accuracy/pred (keys not ignored), accuracy/<key> (only keys with at least one valid label)
  :88-104  distance/<key> for the non-ignored keys with at least one valid label
"""
from typing import Dict, List, Optional

import numpy as np


def evaluate(logits: Dict[str, np.ndarray], labels: np.ndarray, *, shift: bool = True, label_pad_token_id: int = -100,
             ignore_keys: Optional[List[str]] = None, token_values: Optional[Dict[str, np.ndarray]] = None,
             weighted_distance: bool = False) -> Dict[str, float]:
    """logits[key]: float [b, t, V_key] in head order; labels: int [b, t(+1), K]."""
    if shift:
        labels = labels[:, 1:]
    keys = list(logits)
    preds = np.stack([np.argmax(logits[k], axis=-1) for k in keys], axis=-1)
    valid = labels != label_pad_token_id

    def acc(cols):
        m = valid[..., cols]
        return float((preds[..., cols][m] == labels[..., cols][m]).astype(np.float32).mean())

    out = {"accuracy": acc(list(range(len(keys))))}
    if ignore_keys:
        out["accuracy/pred"] = acc([i for i, k in enumerate(keys) if k not in ignore_keys])
    for i, k in enumerate(keys):
        if ignore_keys and k in ignore_keys:
            continue
        if valid[..., i].any():
            out[f"accuracy/{k}"] = acc([i])
    if token_values is not None:
        for i, k in enumerate(keys):
            if (ignore_keys and k in ignore_keys) or not valid[..., i].any():
                continue
            tv = np.asarray(token_values[k], np.float64)
            m = valid[..., i]
            target = tv[labels[..., i][m]]
            if weighted_distance:
                lg = logits[k][m].astype(np.float64)
                p = np.exp(lg - lg.max(-1, keepdims=True))
                p /= p.sum(-1, keepdims=True)
                out[f"distance/{k}"] = float((np.abs(target[:, None] - tv[None, :]) * p).sum(1).mean())
            else:
                out[f"distance/{k}"] = float(np.abs(tv[preds[..., i][m]] - target).mean())
    return out
