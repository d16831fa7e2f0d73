"""Language-modelling wrappers with the reference's contract (`models/scoreperformer/wrappers.py:21-444`)."""
import warnings
from dataclasses import dataclass
from typing import Optional, Dict, Callable, List

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from ...modules.sampling import top_k, filter_logits_and_sample, is_greedy
from ...utils import ExplicitEnum, exists
from .embeddings import shared_tables
from .transformer import TupleTransformer, TupleTransformerOutput, TupleTransformerCaches


class LMWrapper(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self.max_seq_len = self.model.max_seq_len

    def forward(self, seq, labels=None, **kwargs):
        ...


@dataclass
class ScorePerformerLMOutput(TupleTransformerOutput):
    loss: Optional[Tensor] = None
    losses: Optional[Dict[str, Tensor]] = None


_inputs_ready = {"event": None}
_side_streams = {}


def mark_inputs_ready():
    """Called by the model when its forward starts: everything enqueued so far (i.e. the previous step) precedes this point,
    the batch tensors are complete.  Lets the label count below run beside the forward instead of behind it."""
    ev = torch.cuda.Event()
    ev.record()
    _inputs_ready["event"] = ev


def _count_labels_async(labels: Tensor, ignore_index: int):
    """Valid-label count per key, on a side stream, copied to pinned memory: (buffer, event).  The LM head reads it on the
    host to skip keys that carry no loss at all (wrappers.py:56) -- by then the copy finished long ago, so neither the host
    nor the GPU waits."""
    dev = labels.device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    ready = _inputs_ready["event"]
    if ready is None:
        ready = torch.cuda.Event()
        ready.record()
    _inputs_ready["event"] = None
    side.wait_event(ready)
    with torch.cuda.stream(side):
        cnt = (labels != ignore_index).sum(dim=(0, 1), dtype=torch.int32)
        buf = torch.empty(cnt.shape, dtype=torch.int32, pin_memory=True)
        buf.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    labels.record_stream(side)
    return buf, ev


def finalize_lm_losses(out, host_counts: Optional[List[float]] = None):
    """Drop the per-key losses without any valid label, like the `torch.any(labels != ignore)` guard of
    `wrappers.py:56`, given the per-key valid counts on the host; also lets backward skip those keys."""
    keys = list(out._ce_keys)
    if host_counts is None:
        host_counts = torch.stack([out.ce_sums[k][1] for k in keys]).tolist()   # the forward's one host sync
    active = [k for k, c in zip(keys, host_counts) if c > 0]
    out.losses = {k: v for k, v in out.losses.items() if k in active or k not in keys}
    if out._ce_state is not None:
        out._ce_state["active"] = set(active)
    return out


class ScorePerformerLMWrapper(LMWrapper):
    def __init__(self, model: TupleTransformer, ignore_index: int = -100):
        super().__init__(model=model)
        self.ignore_index = ignore_index

    def forward(self, seq: Tensor, labels: Optional[Tensor] = None, _defer_sync: bool = False, **kwargs):
        if exists(labels) and labels.is_cuda and labels.ndim == 3:
            kwargs["label_counts"] = _count_labels_async(labels, self.ignore_index)
        out = self.model(seq, labels=labels, **kwargs)
        loss = losses = None
        res = ScorePerformerLMOutput(loss=None, losses=None, **out.__dict__)
        if exists(labels) and out.ce_sums:
            # per key: mean CE over valid labels; loss = mean over keys that have any valid label (wrappers.py:49-59),
            # computed on device (no sync): keys without labels get weight 0
            keys = list(out.ce_sums.keys())
            sums = torch.stack([out.ce_sums[k] for k in keys])           # [K, 2] (loss sum, valid count)
            has = (sums[:, 1] > 0).float()
            per_key = sums[:, 0] / sums[:, 1].clamp_min(1.0)
            loss = (per_key * has).sum() / has.sum().clamp_min(1.0)
            losses = {k: per_key[i] for i, k in enumerate(keys)}
            res.loss, res.losses = loss, losses
            res._ce_keys, res._ce_state = keys, getattr(self.model.lm_head, "ce_state", None)
            if exists(out.reg_values) and self.model.token_emb.continuous:
                raise NotImplementedError("regression head loss (disabled in every shipped recipe) is not implemented")
            if not _defer_sync:
                finalize_lm_losses(res)
        return res


class ScorePerformerMLMWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, mask_token_id: int = 1, num_special_tokens: int = 4, ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.mask_token_id = mask_token_id
        self.num_special_tokens = num_special_tokens

    @torch.inference_mode()
    def unmask_tokens(self, tokens: Tensor, single_run: bool = True, temperature: float = 1., filter_logits_fn: Callable = top_k,
                      filter_kwargs: Optional[Dict[str, object]] = None, filter_key_ids: Optional[Dict[str, list]] = None,
                      disable_tqdm: bool = False, **kwargs):
        assert callable(filter_logits_fn)
        was_training = self.model.training
        if was_training:
            self.model.eval()
        num_dims = len(tokens.shape)
        if num_dims == 2:
            tokens = tokens[None, :]
        out = tokens.clone().detach()
        mask = kwargs.pop('mask', None)
        if mask is None:
            mask = torch.full_like(out[..., 0], True, dtype=torch.bool, device=out.device)
        filter_key_ids = filter_key_ids or dict()
        unmask_mask = out == self.mask_token_id
        with shared_tables():
            if single_run:
                warnings.warn("`single_run` unmasking with sampling is not yet implemented, using argmax.")
                outputs = self.model(out, mask=mask, **kwargs)
                samples = torch.cat([l.argmax(dim=-1, keepdim=True) for l in outputs.logits.values()], dim=-1)
                out[unmask_mask] = samples[unmask_mask]
            else:
                unmask_ids = torch.where(torch.any(unmask_mask, dim=2))[1]
                for idx in unmask_ids.tolist():
                    type_mask = unmask_mask[:, idx][0]
                    logits_keys = torch.where(type_mask)[0].tolist()
                    outputs = self(out[:, :idx + 1], mask=mask[:, :idx + 1], return_embeddings=True, **kwargs)
                    logits = self.model.lm_head(outputs.hidden_state[:, idx - 1], keys=logits_keys)
                    samples = []
                    for key, logits_i in logits.items():
                        logits_i = logits_i.clone()
                        logits_i[:, :self.num_special_tokens] = -float("Inf")
                        ids = filter_key_ids.get(key, None)
                        if ids is not None:
                            logits_i[:, ids] = -float("Inf")
                        samples.append(_sample(logits_i, filter_logits_fn, filter_kwargs, temperature))
                    out[:, idx, type_mask] = torch.cat(samples, dim=-1)
        if num_dims == 2:
            out = out.squeeze(0)
        if was_training:
            self.model.train(was_training)
        return out


def _sample(logits_i, filter_logits_fn, filter_kwargs, temperature):
    if is_greedy(filter_logits_fn, filter_kwargs):  # top-1 filter + multinomial over a one-hot == argmax
        return logits_i.argmax(dim=-1, keepdim=True)
    return filter_logits_and_sample(logits_i, filter_logits_fn, filter_kwargs=filter_kwargs, temperature=temperature)


class ScorePerformerARWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, pad_token_id: int = 0, eos_token_id: int = 3, num_special_tokens: int = 4,
                 ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.pad_token_id, self.eos_token_id, self.num_special_tokens = pad_token_id, eos_token_id, num_special_tokens

    @torch.inference_mode()
    def generate(self, start_tokens: Tensor, seq_len: int, max_bar: Optional[int] = None, temperature: float = 1.,
                 filter_logits_fn: Callable = top_k, filter_kwargs: Optional[Dict[str, object]] = None,
                 caches: Optional[TupleTransformerCaches] = None, return_caches: bool = False, tokenizer=None,
                 fix_errors: bool = True, disable_tqdm: bool = False, **kwargs):
        assert callable(filter_logits_fn)
        was_training = self.model.training
        if was_training:
            self.model.eval()
        num_dims = len(start_tokens.shape)
        if num_dims == 2:
            start_tokens = start_tokens[None, :]
        b, t = start_tokens.shape[:2]
        out = start_tokens
        mask = kwargs.pop('mask', None)
        if mask is None:
            mask = torch.full_like(out[..., 0], True, dtype=torch.bool, device=out.device)
        with shared_tables():
            for _ in range(t, seq_len + 1):
                x = out[:, -self.max_seq_len:]
                mask = mask[:, -self.max_seq_len:]
                # NOTE: like the reference, `self(...)` applies the CLM shift (forward drops the last position)
                outputs = self(x, mask=mask, caches=caches, return_embeddings=True, return_caches=True, **kwargs)
                logits = self.model.lm_head(outputs.hidden_state[:, -1])
                caches = outputs.caches
                samples = {}
                for key, logits_i in logits.items():
                    logits_i = logits_i.clone()
                    do_sample = True
                    if fix_errors and exists(tokenizer):
                        if key == 'Bar':
                            last_bar = out[:, -1, tokenizer.vocab_types_idx['Bar']]
                            logits_i[:, 4:last_bar] = -float("Inf")
                        same_bar = samples.get('Bar', -1) == out[:, -1, tokenizer.vocab_types_idx['Bar']]
                        if (key == 'Tempo' and same_bar) or key == 'TimeSig':
                            sample = out[:, -1, tokenizer.vocab_types_idx[key]][None]
                            do_sample = False
                    if do_sample:
                        logits_i[:, :2] = -float("Inf")
                        sample = _sample(logits_i, filter_logits_fn, filter_kwargs, temperature)
                    samples[key] = sample
                samples = torch.cat(list(samples.values()), dim=-1)[None]
                out = torch.cat((out, samples), dim=1)
                mask = F.pad(mask, (0, 1), value=True)
                if exists(self.eos_token_id):
                    if (out[..., -1, 0] == self.eos_token_id).any(dim=-1):
                        out[:, -1, 1:] = self.pad_token_id
                        break
                elif exists(max_bar):
                    if (out[..., -1, 0] > max_bar).any(dim=-1):
                        out = out[:, :-1, :]
                        break
        out = out[:, t:]
        if num_dims == 2:
            out = out.squeeze(0)
        if was_training:
            self.model.train(was_training)
        if return_caches:
            return out, caches
        return out

    def forward(self, seq: Tensor, labels: Optional[Tensor] = None, **kwargs):
        seq = seq[:, :-1]
        labels = labels[:, 1:] if exists(labels) else None
        context = kwargs.get("context", None)
        if exists(context) and self.model.context_emb_mode == "cat":
            kwargs["context"] = context[:, 1:]
        style_embeddings = kwargs.get("style_embeddings", None)
        if exists(style_embeddings):
            kwargs["style_embeddings"] = style_embeddings[:, 1:]
        mask = kwargs.get('mask', None)
        if exists(mask) and mask.shape[1] == seq.shape[1] + 1:
            kwargs['mask'] = mask[:, :-1]
        return super().forward(seq, labels=labels, **kwargs)


class ScorePerformerMixedLMWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, pad_token_id: int = 0, mask_token_id: int = 1, num_special_tokens: int = 4,
                 ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.pad_token_id, self.mask_token_id, self.num_special_tokens = pad_token_id, mask_token_id, num_special_tokens

    @torch.inference_mode()
    def unmask_tokens(self, tokens: Tensor, tokens_masked, temperature: float = 1., filter_logits_fn: Callable = top_k,
                      filter_kwargs: Optional[Dict[str, object]] = None, filter_key_ids: Optional[Dict[str, list]] = None,
                      caches: Optional[TupleTransformerCaches] = None, return_caches: bool = False, disable_tqdm: bool = False,
                      **kwargs):
        assert callable(filter_logits_fn)
        was_training = self.model.training
        if was_training:
            self.model.eval()
        num_dims = len(tokens.shape)
        if num_dims == 2:
            tokens = tokens[None, :]
            tokens_masked = tokens_masked[None, :]
        out = tokens.clone().detach()
        mask = kwargs.pop('mask', None)
        if mask is None:
            mask = torch.full_like(out[..., 0], True, dtype=torch.bool, device=out.device)
        # fast path: greedy, one sequence, fresh caches -> hipGraph-replayed fp32 decode engine (decode.py)
        if (out.is_cuda and out.shape[0] == 1 and caches is None and not filter_key_ids and is_greedy(filter_logits_fn, filter_kwargs)
                and bool(mask.all()) and getattr(self, "use_decode_engine", True)):
            try:
                from ...decode import GreedyDecoder
                eng = GreedyDecoder(self.model, out.shape[1])
            except NotImplementedError:
                eng = None
            if eng is not None:
                res, _ = eng.run(out, tokens_masked, kwargs.get("context"), kwargs.get("style_embeddings"), self.mask_token_id)
                res = res.squeeze(0) if num_dims == 2 else res
                if was_training:
                    self.model.train(was_training)
                return (res, eng.caches()) if return_caches else res
        filter_key_ids = filter_key_ids or dict()
        unmask_mask = out == self.mask_token_id
        # one host read of the mask layout for the whole window (the reference reads it per step: wrappers.py:385-390)
        unmask_host = unmask_mask[0].cpu()
        unmask_ids = torch.where(torch.any(unmask_host, dim=1))[0].tolist()
        with shared_tables():
            for idx in unmask_ids:
                type_mask = unmask_mask[:, idx][0]
                logits_keys = torch.where(unmask_host[idx])[0].tolist()
                outputs = self(out[:, :idx + 1], seq_masked=tokens_masked[:, :idx + 1], mask=mask[:, :idx + 1],
                               return_embeddings=True, return_caches=True, caches=caches, **kwargs)
                caches = outputs.caches
                logits = self.model.lm_head(outputs.hidden_state[:, idx - 1], keys=logits_keys)
                samples = []
                for key, logits_i in logits.items():
                    logits_i = logits_i.clone()
                    logits_i[:, self.pad_token_id] = -float("Inf")
                    logits_i[:, self.mask_token_id] = -float("Inf")
                    ids = filter_key_ids.get(key, None)
                    if ids is not None:
                        logits_i[:, ids] = -float("Inf")
                    samples.append(_sample(logits_i, filter_logits_fn, filter_kwargs, temperature))
                out[:, idx, type_mask] = torch.cat(samples, dim=-1)
        if num_dims == 2:
            out = out.squeeze(0)
        if was_training:
            self.model.train(was_training)
        if return_caches:
            return out, caches
        return out

    def forward(self, seq: Tensor, labels: Optional[Tensor] = None, **kwargs):
        seq = seq[:, :-1]
        labels = labels[:, 1:] if exists(labels) else None
        seq_masked = kwargs.pop("seq_masked", None)
        if exists(seq_masked):
            seq_masked = seq_masked[:, 1:]
        context = kwargs.get("context", None)
        if exists(context) and self.model.context_emb_mode == "cat":
            kwargs["context"] = context[:, 1:]
        style_embeddings = kwargs.get("style_embeddings", None)
        if exists(style_embeddings):
            kwargs["style_embeddings"] = style_embeddings[:, 1:]
        mask = kwargs.get("mask", None)
        if exists(mask) and mask.shape[1] == seq.shape[1] + 1:
            kwargs["mask"] = mask[:, :-1]
        return super().forward(seq, labels=labels, x_extra=seq_masked, **kwargs)


class ScorePerformerLMModes(ExplicitEnum):
    MLM = "mlm"
    CLM = "clm"
    MixedLM = "mixlm"


ScorePerformerLMWrappers = {
    ScorePerformerLMModes.MLM: ScorePerformerMLMWrapper,
    ScorePerformerLMModes.CLM: ScorePerformerARWrapper,
    ScorePerformerLMModes.MixedLM: ScorePerformerMixedLMWrapper
}
