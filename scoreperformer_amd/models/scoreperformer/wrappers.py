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
from ...utils.amp import no_autocast


class LMWrapper(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self.max_seq_len = self.model.max_seq_len

    @no_autocast
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

    @no_autocast
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
            if exists(out.reg_values) and self.model.token_emb.continuous:
                reg = self._regression_losses(out, labels)
                if reg:
                    loss = loss + sum(reg.values()) / len(reg)
                    losses.update(reg)
            res.loss, res.losses = loss, losses
            res._ce_keys, res._ce_state = keys, getattr(self.model.lm_head, "ce_state", None)
            if not _defer_sync:
                finalize_lm_losses(res)
        return res

    def _regression_losses(self, out, labels: Tensor) -> Dict[str, Tensor]:
        """L1 distance between the regression head's scalar and the label's normalised token value, over non-special labels
        (`wrappers.py:61-78`; the head is disabled in every shipped recipe, so this is device-side glue, not a kernel).  The reference
        gathers the selected rows with a boolean mask (a host sync); the masked mean below is the same number without one."""
        reg = {}
        for i, key in enumerate(out.logits.keys()):      # the reference indexes the label column by the position in `logits`
            values = out.reg_values.get(key)
            if values is None:
                continue
            column = labels[..., i]
            chosen = column > 3                           # special ids carry no value
            targets = F.embedding(column.clamp_min(0), self.model.token_emb.embs[key].token_values)
            err = (values.float() - targets).abs().squeeze(-1) * chosen
            reg[f"{key}/l1"] = err.sum() / chosen.sum()
        return reg


class _decode_call:
    """Shared frame of the decode entry points (`unmask_tokens`, `generate`): run the wrapped transformer in eval mode and hand the
    caller back whatever mode it was in; token arrays given without a batch axis get one for the duration of the call."""

    def __init__(self, wrapper: LMWrapper, *token_arrays: Tensor, mask: Optional[Tensor] = None):
        self.net = wrapper.model
        self.resume_training = self.net.training
        self.unbatched = token_arrays[0].ndim == 2
        self.arrays = [t.unsqueeze(0) if self.unbatched else t for t in token_arrays]
        self.mask = mask

    def __enter__(self):
        if self.resume_training:
            self.net.eval()
        if self.mask is None:   # every position is a real note
            first = self.arrays[0]
            self.mask = torch.ones(first.shape[:2], dtype=torch.bool, device=first.device)
        return self

    def __exit__(self, *exc):
        if self.resume_training:
            self.net.train()
        return False

    def result(self, tokens: Tensor) -> Tensor:
        return tokens[0] if self.unbatched else tokens


def contiguous_note_spans(mask: Tensor):
    """[(first, end)] per sequence when the True entries of every row of `mask` [b, n] form ONE non-empty contiguous block (un-padded,
    right-padded, front-padded or both), else None (a hole, or an empty row).  One host read."""
    mi = mask.int()
    edges = mi[:, 1:] - mi[:, :-1]
    rising = (edges == 1).sum(1) + mi[:, 0]          # blocks per row: a block starts at position 0 or at a rising edge
    if not bool((rising == 1).all()):
        return None
    first = torch.argmax(mi, dim=1)
    count = mi.sum(dim=1)
    return [(int(a), int(a) + int(c)) for a, c in zip(first.tolist(), count.tolist())]


def _forbid(logits: Tensor, ids) -> Tensor:
    """A copy of `logits` [rows, V] in which the given ids (slice, list or int) can never be drawn."""
    logits = logits.clone()
    if ids is not None:
        logits[:, ids] = -float("Inf")
    return logits


def _sample(logits_i, filter_logits_fn, filter_kwargs, temperature):
    if is_greedy(filter_logits_fn, filter_kwargs):  # top-1 filter + multinomial over a one-hot == argmax
        return logits_i.argmax(dim=-1, keepdim=True)
    return filter_logits_and_sample(logits_i, filter_logits_fn, filter_kwargs=filter_kwargs, temperature=temperature)


def _teacher_forcing_views(net: TupleTransformer, seq: Tensor, labels: Optional[Tensor], kwargs: dict):
    """Next-note prediction: position t of the input predicts note t + 1, so inputs lose their last position and everything that
    describes the TARGET note -- labels, per-note style embeddings, a channel-concatenated score context -- loses its first
    (`wrappers.py:290-307,409-431`).  A context that is cross-attended is a whole sequence and stays as it is.  Edits `kwargs`."""
    inputs = seq[:, :-1]
    targets = labels[:, 1:] if exists(labels) else None
    for name in ("context", "style_embeddings"):
        cond = kwargs.get(name)
        if exists(cond) and (name != "context" or net.context_emb_mode == "cat"):
            kwargs[name] = cond[:, 1:]
    mask = kwargs.get("mask")
    if exists(mask) and mask.shape[1] == inputs.shape[1] + 1:
        kwargs["mask"] = mask[:, :-1]
    return inputs, targets


class ScorePerformerMLMWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, mask_token_id: int = 1, num_special_tokens: int = 4, ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.mask_token_id = mask_token_id
        self.num_special_tokens = num_special_tokens

    @torch.inference_mode()
    @no_autocast
    def unmask_tokens(self, tokens: Tensor, single_run: bool = True, temperature: float = 1., filter_logits_fn: Callable = top_k,
                      filter_kwargs: Optional[Dict[str, object]] = None, filter_key_ids: Optional[Dict[str, list]] = None,
                      disable_tqdm: bool = False, **kwargs):
        """Fill every MASK sub-token (`wrappers.py:100-182`): in one pass by arg-max, or note by note with the chosen sampler."""
        assert callable(filter_logits_fn)
        banned = filter_key_ids or {}
        with _decode_call(self, tokens, mask=kwargs.pop("mask", None)) as call, shared_tables():
            filled = call.arrays[0].clone().detach()
            holes = filled == self.mask_token_id
            if single_run:
                warnings.warn("`single_run` unmasking with sampling is not yet implemented, using argmax.")
                logits = self.model(filled, mask=call.mask, **kwargs).logits
                best = torch.stack([l.argmax(dim=-1) for l in logits.values()], dim=-1)
                filled[holes] = best[holes]
            else:
                for idx in torch.where(holes.any(dim=2))[1].tolist():
                    dims = holes[0, idx]
                    hidden = self(filled[:, :idx + 1], mask=call.mask[:, :idx + 1], return_embeddings=True, **kwargs).hidden_state
                    logits = self.model.lm_head(hidden[:, idx - 1], keys=torch.where(dims)[0].tolist())
                    draws = []
                    for key, logits_k in logits.items():
                        logits_k = _forbid(_forbid(logits_k, slice(0, self.num_special_tokens)), banned.get(key))
                        draws.append(_sample(logits_k, filter_logits_fn, filter_kwargs, temperature))
                    filled[:, idx, dims] = torch.cat(draws, dim=-1)
            return call.result(filled)


class ScorePerformerARWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, pad_token_id: int = 0, eos_token_id: int = 3, num_special_tokens: int = 4,
                 ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.pad_token_id, self.eos_token_id, self.num_special_tokens = pad_token_id, eos_token_id, num_special_tokens

    def _musical_constraint(self, key: str, logits_k: Tensor, previous: Tensor, drawn: Dict[str, Tensor], tokenizer):
        """The reference's three repairs of a sampled note (`wrappers.py:246-259`): bars never run backwards; the tempo is frozen
        inside a bar; the time signature is copied from the previous note.  Returns (forced token or None, logits to sample from)."""
        column = tokenizer.vocab_types_idx
        if key == "Bar":
            earliest = previous[:, column["Bar"]]
            return None, _forbid(logits_k, slice(4, earliest))
        bar_unchanged = drawn.get("Bar", -1) == previous[:, column["Bar"]]
        if key == "TimeSig" or (key == "Tempo" and bar_unchanged):
            return previous[:, column[key]][None], logits_k
        return None, logits_k

    @torch.inference_mode()
    @no_autocast
    def generate(self, start_tokens: Tensor, seq_len: int, max_bar: Optional[int] = None, temperature: float = 1.,
                 filter_logits_fn: Callable = top_k, filter_kwargs: Optional[Dict[str, object]] = None,
                 caches: Optional[TupleTransformerCaches] = None, return_caches: bool = False, tokenizer=None,
                 fix_errors: bool = True, disable_tqdm: bool = False, **kwargs):
        """Autoregressive continuation with key/value caches (`wrappers.py:200-288`)."""
        assert callable(filter_logits_fn)
        repair = fix_errors and exists(tokenizer)
        with _decode_call(self, start_tokens, mask=kwargs.pop("mask", None)) as call, shared_tables():
            notes, mask = call.arrays[0], call.mask
            prompt_len = notes.shape[1]
            for _ in range(prompt_len, seq_len + 1):
                window, mask = notes[:, -self.max_seq_len:], mask[:, -self.max_seq_len:]
                # like the reference, `self(...)` applies the CLM shift: forward drops the last position
                step = self(window, mask=mask, caches=caches, return_embeddings=True, return_caches=True, **kwargs)
                caches = step.caches
                drawn: Dict[str, Tensor] = {}
                for key, logits_k in self.model.lm_head(step.hidden_state[:, -1]).items():
                    forced = None
                    if repair:
                        forced, logits_k = self._musical_constraint(key, logits_k, notes[:, -1], drawn, tokenizer)
                    if forced is None:
                        forced = _sample(_forbid(logits_k, slice(0, 2)), filter_logits_fn, filter_kwargs, temperature)
                    drawn[key] = forced
                notes = torch.cat((notes, torch.cat(list(drawn.values()), dim=-1)[None]), dim=1)
                mask = F.pad(mask, (0, 1), value=True)
                newest_first_dim = notes[..., -1, 0]
                if exists(self.eos_token_id):
                    if (newest_first_dim == self.eos_token_id).any(dim=-1):
                        notes[:, -1, 1:] = self.pad_token_id      # an end-of-sequence note carries nothing else
                        break
                elif exists(max_bar) and (newest_first_dim > max_bar).any(dim=-1):
                    notes = notes[:, :-1, :]                      # the note that opened a bar beyond the limit is dropped
                    break
            continuation = call.result(notes[:, prompt_len:])
        return (continuation, caches) if return_caches else continuation

    @no_autocast
    def forward(self, seq: Tensor, labels: Optional[Tensor] = None, **kwargs):
        inputs, targets = _teacher_forcing_views(self.model, seq, labels, kwargs)
        return super().forward(inputs, labels=targets, **kwargs)


class ScorePerformerMixedLMWrapper(ScorePerformerLMWrapper):
    def __init__(self, model: TupleTransformer, pad_token_id: int = 0, mask_token_id: int = 1, num_special_tokens: int = 4,
                 ignore_index: int = -100):
        super().__init__(model=model, ignore_index=ignore_index)
        self.pad_token_id, self.mask_token_id, self.num_special_tokens = pad_token_id, mask_token_id, num_special_tokens

    def _engine_for(self, filled: Tensor, mask: Tensor, caches, banned, filter_logits_fn, filter_kwargs):
        """The hipGraph-replayed fp32 decode engine (decode.py) serves the common call: greedy, fresh caches -- one sequence, or (round 5)
        several with the SAME layout of MASK sub-tokens, which is what the reference's loop assumes anyway (it takes the layout of batch
        element 0 for all, wrappers.py:385-396): the sequences then go through the engine one after the other.  PADDED batches too, when
        every sequence's notes are one contiguous block (right-padded, front-padded or both): a causal decoder's valid positions never
        see the padded keys behind them, masked keys in front of them are invisible, and every position-dependent term (ALiBi, the
        per-note context / style) is relative or per note -- so every sequence runs over its own valid block and the padded positions are
        left as given (the reference writes draws from padded rows there: don't-care values).  Holed masks take the module path.  Returns
        (engine, per-sequence (first, end) spans or None); (None, None) = module path."""
        usable = (filled.is_cuda and caches is None and not banned
                  and is_greedy(filter_logits_fn, filter_kwargs) and getattr(self, "use_decode_engine", True))
        lens = None
        if usable and not bool(mask.all()):
            m = mask.bool()
            spans = contiguous_note_spans(m)
            usable = spans is not None
            if usable:
                first = torch.tensor([a for a, _ in spans], device=m.device)
                holes = (filled == self.mask_token_id) & m[..., None]
                # one MASK layout on every valid position, nothing to decode where element 0 (whose layout the reference uses) is padded,
                # and nothing to decode at a block's own first note behind padding (it has no valid predecessor to be predicted from)
                at_first = holes[torch.arange(holes.shape[0], device=holes.device), first].any(dim=1) & (first > 0)
                usable = (bool(((holes == holes[:1]) | ~m[..., None]).all()) and not bool((holes & ~m[:1, :, None]).any())
                          and not bool(at_first.any()))
                lens = spans
        elif usable and filled.shape[0] > 1:
            holes = filled == self.mask_token_id
            usable = bool((holes == holes[:1]).all())
        if not usable:
            return None, None
        try:
            from ...decode import GreedyDecoder
            # `reference_compat` (attribute, default False): cross-attending decoders reproduce the reference's stale-hidden-row behaviour
            # token for token (decode.GreedyDecoder; wrappers.py:364 with modules/transformer/transformer.py:201)
            return GreedyDecoder(self.model, filled.shape[1], reference_compat=bool(getattr(self, "reference_compat", False))), lens
        except NotImplementedError:
            return None, None

    @staticmethod
    def _stack_caches(per_seq, spans=None):
        """Caches of single-sequence engine runs -> one TupleTransformerCaches with the batch in front (the reference's layout).  Runs over
        blocks of different extents (a padded batch; `spans` = their (first, end) positions) are zero-padded along the position axis: in
        front up to the block's first position, behind up to the longest end."""
        from .transformer import TupleTransformerCaches
        from ...modules.transformer.attend import AttentionIntermediates
        from ...modules.transformer.transformer import TransformerIntermediates
        spans = spans or [(0, None)] * len(per_seq)

        def cat(ts, pos_dim):
            # a cache tensor of run i covers positions first_i .. first_i + its own length (the shifted input drops the last note)
            n = max(a + t.shape[pos_dim] for t, (a, _) in zip(ts, spans))
            out = []
            for t, (a, _) in zip(ts, spans):
                back = n - a - t.shape[pos_dim]
                out.append(t if a == 0 and back == 0 else F.pad(t, (0, 0) * (t.ndim - 1 - pos_dim) + (a, back)))
            return torch.cat(out, dim=0)

        first = per_seq[0]
        hid = [cat([c.transformer.hiddens[i] for c in per_seq], 1) for i in range(len(first.transformer.hiddens))]
        att = []
        for i in range(len(first.transformer.attention)):
            k0 = first.transformer.attention[i].keys
            pd = 1 if k0.ndim == 3 else 2            # [1, n, 64] multi-query, [1, h, n, 64] otherwise
            att.append(AttentionIntermediates(keys=cat([c.transformer.attention[i].keys for c in per_seq], pd),
                                              values=cat([c.transformer.attention[i].values for c in per_seq], pd)))
        return TupleTransformerCaches(token_emb=cat([c.token_emb for c in per_seq], 1),
                                      transformer=TransformerIntermediates(hiddens=hid, attention=att))

    @torch.inference_mode()
    @no_autocast
    def unmask_tokens(self, tokens: Tensor, tokens_masked, temperature: float = 1., filter_logits_fn: Callable = top_k,
                      filter_kwargs: Optional[Dict[str, object]] = None, filter_key_ids: Optional[Dict[str, list]] = None,
                      caches: Optional[TupleTransformerCaches] = None, return_caches: bool = False, disable_tqdm: bool = False,
                      **kwargs):
        """Cached note-by-note rendering of the MASK sub-tokens (`wrappers.py:325-407`): the decode hot loop."""
        assert callable(filter_logits_fn)
        banned = filter_key_ids or {}
        with _decode_call(self, tokens, tokens_masked, mask=kwargs.pop("mask", None)) as call:
            filled, masked_view = call.arrays[0].clone().detach(), call.arrays[1]
            engine, lens = self._engine_for(filled, call.mask, caches, banned, filter_logits_fn, filter_kwargs)
            if engine is not None and filled.shape[0] == 1 and lens is None:
                filled, _ = engine.run(filled, masked_view, kwargs.get("context"), kwargs.get("style_embeddings"), self.mask_token_id,
                                       context_mask=kwargs.get("context_mask"))
                caches = engine.caches() if return_caches else None
            elif engine is not None:       # several sequences with one MASK layout (each over its valid prefix): one engine run each
                L = filled.shape[1]
                per_note_ctx = self.model.context_emb_mode == "cat"      # else: a whole cross-attended sequence with its own mask

                def pick(t, i, a, e, per_note=True):
                    return None if t is None else (t[i:i + 1, a:e] if per_note else t[i:i + 1])

                rows, per_seq = [], []
                for i in range(filled.shape[0]):
                    a, e = (0, L) if lens is None else lens[i]
                    row, _ = engine.run(filled[i:i + 1, a:e], masked_view[i:i + 1, a:e], pick(kwargs.get("context"), i, a, e, per_note_ctx),
                                        pick(kwargs.get("style_embeddings"), i, a, e), self.mask_token_id,
                                        context_mask=pick(kwargs.get("context_mask"), i, a, e, False))
                    rows.append(row.clone() if (a, e) == (0, L) else torch.cat([filled[i:i + 1, :a], row, filled[i:i + 1, e:]], dim=1))
                    if return_caches:    # the engine's buffers are overwritten by the next sequence
                        c = engine.caches()
                        per_seq.append(type(c)(token_emb=c.token_emb.clone(), transformer=type(c.transformer)(
                            hiddens=[h.clone() for h in c.transformer.hiddens],
                            attention=[type(a)(keys=a.keys.clone(), values=a.values.clone()) for a in c.transformer.attention])))
                filled = torch.cat(rows, dim=0)
                caches = self._stack_caches(per_seq, lens) if return_caches else None
            else:
                holes = filled == self.mask_token_id
                holes_host = holes[0].cpu()   # ONE host read of the mask layout per window (the reference reads it per note: :385-390)
                with shared_tables():
                    for idx in torch.where(holes_host.any(dim=1))[0].tolist():
                        step = self(filled[:, :idx + 1], seq_masked=masked_view[:, :idx + 1], mask=call.mask[:, :idx + 1],
                                    return_embeddings=True, return_caches=True, caches=caches, **kwargs)
                        caches = step.caches
                        logits = self.model.lm_head(step.hidden_state[:, idx - 1], keys=torch.where(holes_host[idx])[0].tolist())
                        draws = []
                        for key, logits_k in logits.items():
                            logits_k = _forbid(_forbid(logits_k, [self.pad_token_id, self.mask_token_id]), banned.get(key))
                            draws.append(_sample(logits_k, filter_logits_fn, filter_kwargs, temperature))
                        filled[:, idx, holes[0, idx]] = torch.cat(draws, dim=-1)
            filled = call.result(filled)
        return (filled, caches) if return_caches else filled

    @no_autocast
    def forward(self, seq: Tensor, labels: Optional[Tensor] = None, **kwargs):
        masked_view = kwargs.pop("seq_masked", None)
        inputs, targets = _teacher_forcing_views(self.model, seq, labels, kwargs)
        return super().forward(inputs, labels=targets, x_extra=masked_view[:, 1:] if exists(masked_view) else None, **kwargs)


class ScorePerformerLMModes(ExplicitEnum):
    MLM = "mlm"
    CLM = "clm"
    MixedLM = "mixlm"


ScorePerformerLMWrappers = {
    ScorePerformerLMModes.MLM: ScorePerformerMLMWrapper,
    ScorePerformerLMModes.CLM: ScorePerformerARWrapper,
    ScorePerformerLMModes.MixedLM: ScorePerformerMixedLMWrapper
}
