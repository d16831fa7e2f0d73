"""Per-step metrics with the reference's contract (`models/scoreperformer/evaluator.py:15-106`).

Uses the arg-max that the fused head/cross-entropy kernel already produced when available (`outputs.argmax`), so the
logits are not read a second time.
"""
from typing import Optional, List

import torch
import torch.nn.functional as F

from .wrappers import ScorePerformerLMModes


class ScorePerformerEvaluator:
    def __init__(self, model, tokenizer=None, label_pad_token_id: int = -100, weighted_distance: bool = False,
                 ignore_keys: Optional[List[str]] = None, token_values=None):
        self.model = model
        self.tokenizer = tokenizer
        self.label_pad_token_id = label_pad_token_id
        self.weighted_distance = weighted_distance
        self.ignore_keys = ignore_keys
        self.token_values = None
        if token_values is not None:
            self.token_values = {k: torch.as_tensor(v, dtype=torch.float32)[:, None] for k, v in token_values.items()}
        elif tokenizer is not None:
            self.token_values = {k: torch.from_numpy(v)[:, None] for k, v in tokenizer.token_values(normalize=False).items()}

    def _accuracy(self, predictions, labels):
        m = labels != self.label_pad_token_id
        return (predictions[m] == labels[m]).float().mean()

    @torch.no_grad()
    def __call__(self, inputs, outputs, ignore_keys: Optional[List[str]] = None):
        metrics = {}
        ignore_keys = ignore_keys or self.ignore_keys
        labels = inputs["labels"] if isinstance(inputs, dict) else inputs.labels.tokens
        if self.model.mode in (ScorePerformerLMModes.CLM, ScorePerformerLMModes.MixedLM):
            labels = labels[:, 1:]
        if hasattr(outputs, "perf_decoder"):
            outputs = outputs.perf_decoder
        labels = labels.to(outputs.hidden_state.device)
        keys = list(outputs.logits.keys())
        am = getattr(outputs, "argmax", None)
        preds = torch.stack([(am[k].long() if am and k in am else outputs.logits[k].argmax(dim=-1)) for k in keys], dim=-1)
        metrics["accuracy"] = self._accuracy(preds, labels)
        if ignore_keys:
            use = torch.tensor([i for i, k in enumerate(keys) if k not in ignore_keys], device=preds.device)
            metrics["accuracy/pred"] = self._accuracy(preds[..., use], labels[..., use])
        valid = (labels != self.label_pad_token_id).flatten(0, -2).any(dim=0).tolist()
        for i, key in enumerate(keys):
            if ignore_keys and key in ignore_keys:
                continue
            if valid[i]:
                metrics[f"accuracy/{key}"] = self._accuracy(preds[..., i], labels[..., i])
        if self.token_values is not None:
            for i, key in enumerate(keys):
                if (ignore_keys and key in ignore_keys) or not valid[i]:
                    continue
                tv = self.token_values[key] = self.token_values[key].to(preds.device)
                m = labels[..., i] != self.label_pad_token_id
                targets = F.embedding(labels[..., i][m], tv)
                if self.weighted_distance:
                    probs = outputs.logits[key].float().softmax(dim=-1)[m]
                    metrics[f"distance/{key}"] = ((targets[:, None] - tv[None, :]).abs() * probs[..., None]).sum(dim=1).mean()
                else:
                    metrics[f"distance/{key}"] = (F.embedding(preds[..., i][m], tv) - targets).abs().float().mean()
        return metrics
