"""Per-step metrics with the reference's contract (`models/scoreperformer/evaluator.py:15-106`).

Fused path: `attach()` hands the token values to the LM head; the cross-entropy kernel of the train step then
also accumulates, per key, #(argmax == label) and the (weighted) token-value distance in the SAME pass over the logits
(`spn_ce_fwd_eval`), and `__call__` only divides device scalars: no second pass over the logits, no softmax tensor, no boolean
gathers, no host sync.  Outputs of a forward that ran without an attached evaluator take the unfused path, which still reuses the
kernel's arg-max when present.
"""
from typing import Optional, List

import torch
import torch.nn.functional as F

from .wrappers import ScorePerformerLMModes
from ...utils.amp import no_autocast


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

    # ---- fused path ---------------------------------------------------------------------------------------------------------
    def _lm_head(self):
        dec = getattr(self.model, "perf_decoder", self.model)
        return getattr(getattr(dec, "model", dec), "lm_head", None)

    def attach(self, device=None):
        """Make the following forwards (with labels) produce the metric sums inside the cross-entropy kernel."""
        head = self._lm_head()
        if head is None:
            raise RuntimeError("the model has no LM head to attach the evaluator to")
        device = device or next(self.model.parameters()).device
        values = None
        if self.token_values is not None:
            values = {k: v.to(device=device, dtype=torch.float32).reshape(-1).contiguous() for k, v in self.token_values.items()}
        head.eval_spec = {"values": values, "weighted": self.weighted_distance}
        return self

    def detach(self):
        head = self._lm_head()
        if head is not None and hasattr(head, "eval_spec"):
            del head.eval_spec

    def _from_sums(self, outputs, ignore_keys):
        keys = list(outputs.ce_sums.keys())
        valid = torch.stack([outputs.ce_sums[k][1] for k in keys])                        # [K] valid-label counts (device)
        zero = valid.new_zeros(2)
        sums = torch.stack([outputs.eval_sums.get(k, zero) for k in keys])               # [K, 2] (#correct, distance sum)
        state = getattr(outputs, "_ce_state", None) or {}
        counts = state.get("counts")                                                       # host copy made before the decoder ran
        if counts is None or len(counts) != len(keys):
            counts = valid.tolist()
        live = [c > 0 for c in counts[:len(keys)]]
        metrics = {"accuracy": sums[:, 0].sum() / valid.sum()}
        if ignore_keys:
            use = torch.tensor([i for i, k in enumerate(keys) if k not in ignore_keys], device=valid.device)
            metrics["accuracy/pred"] = sums[use, 0].sum() / valid[use].sum()
        per_key = sums / valid[:, None]
        for i, key in enumerate(keys):
            if not (ignore_keys and key in ignore_keys) and live[i]:
                metrics[f"accuracy/{key}"] = per_key[i, 0]
        if self.token_values is not None:
            for i, key in enumerate(keys):
                if not (ignore_keys and key in ignore_keys) and live[i]:
                    metrics[f"distance/{key}"] = per_key[i, 1]
        return metrics

    # ---- reference contract --------------------------------------------------------------------------------------------------
    def _accuracy(self, predictions, labels):
        m = labels != self.label_pad_token_id
        return (predictions[m] == labels[m]).float().mean()

    @torch.no_grad()
    @no_autocast
    def __call__(self, inputs, outputs, ignore_keys: Optional[List[str]] = None):
        metrics = {}
        ignore_keys = ignore_keys or self.ignore_keys
        if hasattr(outputs, "perf_decoder"):
            outputs = outputs.perf_decoder
        if getattr(outputs, "eval_sums", None) and getattr(outputs, "ce_sums", None):
            return self._from_sums(outputs, ignore_keys)
        labels = inputs["labels"] if isinstance(inputs, dict) else inputs.labels.tokens
        if self.model.mode in (ScorePerformerLMModes.CLM, ScorePerformerLMModes.MixedLM):
            labels = labels[:, 1:]
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
