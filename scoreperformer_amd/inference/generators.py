"""Render loop around the cached decode path: the reference's `ScorePerformerGenerator`
(`scoreperformer/inference/generators.py:35-443`) with the same constructor, methods, arguments and results.

What changes is where the state lives.  The reference keeps the growing window as device tensors and asks the device for every
control decision (`torch.where(torch.diff(...))` for bar boundaries, chord comparison, `.cpu().numpy()` of each group), re-runs the
decoder modules per note with `torch.cat`-grown caches, slices all caches after every time window and rebuilds them whenever the
context is cropped.  Here the token window is host state (the generated tokens have to reach the host anyway: the MIDI messenger
turns them into onset times), every control decision is numpy on it, and the decoder is a `decode.RenderSession`: static caches,
one captured step replayed per note, cutting the caches = lowering an integer.  One D2H copy of `num_new_notes x K` tokens per
chord group is the only device->host traffic.  Sampling other than greedy, or a decoder the engine does not cover, takes the
module path (`perf_decoder.unmask_tokens` with `TupleTransformerCaches`), like the reference.

The dataset / tokenizer / messenger are the reference's objects (duck-typed here: miditok is not a dependency of this package).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Union

import numpy as np
import torch
from torch import Tensor

from ..models.scoreperformer.transformer import TupleTransformerCaches
from ..models.scoreperformer.wrappers import is_greedy
from ..modules.sampling import top_k
from ..modules.transformer.attend import AttentionIntermediates
from ..modules.transformer.transformer import TransformerIntermediates

SOS_TOKEN, EOS_TOKEN = "SOS_None", "EOS_None"          # data/tokenizers/constants.py:8-9
DEFAULT_TEMPO = 120                                    # miditok.constants.TEMPO


@dataclass
class PerformanceData:                                  # generators.py:24-33
    perf_seq: Optional[np.ndarray] = None
    notes: Optional[Tensor] = None
    embeddings: Optional[Tensor] = None
    context: Optional[Tensor] = None
    gen_seq: Optional[Tensor] = None
    intermediates: Optional[object] = None
    caches: Optional[object] = None                     # TupleTransformerCaches (module path) or the engine's cache tag
    reached_eos: bool = False


def find_closest(array, values):                        # utils/functions.py:41-55
    ids = np.searchsorted(array, values, side="left")
    right = array[np.minimum(ids, len(array) - 1)]
    left = array[np.maximum(ids - 1, 0)]
    return ids - ((ids == len(array)) | (np.fabs(values - left) < np.fabs(values - right)))


def get_end_bar(score_indices, start_bar=0, max_seq_len=512, max_bar=256):   # data/datasets/utils.py:56-58
    end_bar = np.where(score_indices <= score_indices[start_bar] + max_seq_len)[0][-1] - 1
    return min(max(start_bar, end_bar), start_bar + max_bar - 1)


@dataclass
class ScorePerformanceSampleMeta:                       # data/datasets/score_performance.py:37-50 (what `dataset.get(meta=...)` reads)
    idx: Optional[int]
    score_idx: int
    perf_idx: int
    start_bar: int
    end_bar: Optional[int]
    start_idx: Optional[int] = None
    end_idx: Optional[int] = None
    bar_offset: int = 0
    note_shifts: tuple = (0, 0)
    augmentations: Optional[object] = None
    noisy_augmentations: Optional[object] = None
    is_deadpan: bool = False


@dataclass
class _EngineCache:
    """What `perf_data.caches` holds on the engine path: which window the session's cache rows belong to and how many are valid."""
    start_idx: int
    length: int


class ScorePerformerGenerator:
    def __init__(self, model, dataset, collator, messenger, device: Optional[Union[str, torch.device]] = None,
                 use_engine: bool = True, engine_max_len: int = 1024, prefill: str = "engine", prefill_min: int = 32):
        """use_engine: greedy decoding through `decode.RenderSession`.  prefill: how the caches of a window are rebuilt after the
        context was cropped (or on a cold start with a long known prefix): "engine" = one batched pass of the fp32 engine
        (`RenderSession.prefill`: exact-fp32 GEMMs + position-parallel attention; reproduces the reference's fp32 tokens),
        "sequential" = the fp32 engine note by note, "modules" = one batched module forward over the window (bf16 GEMMs; not
        bit-exact in near-tied arg-maxes), adopted by the session when the prefix has at least `prefill_min` rows."""
        self.model = model
        assert model.perf_decoder is not None
        self.dataset = dataset
        self.tokenizer = dataset.tokenizer
        self.collator = collator
        self.sos_token_id = self.tokenizer[0, SOS_TOKEN]
        self.eos_token_id = self.tokenizer[0, EOS_TOKEN]
        self.messenger = messenger
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        if prefill not in ("engine", "sequential", "modules"):
            raise ValueError("prefill must be 'engine', 'sequential' or 'modules'")
        self.use_engine, self.engine_max_len, self.prefill, self.prefill_min = use_engine, engine_max_len, prefill, prefill_min
        self._session = None
        self.sampling_seed = 0
        self._init_variables()
        self.perf_data = PerformanceData()

    def _init_variables(self):
        num_dims = len(self.tokenizer.sizes)
        mask_dims = sorted(set(range(num_dims)).difference(self.collator.mask_ignore_token_dims))
        self.mask_dims = torch.tensor(mask_dims)
        self._mask_dims = np.asarray(mask_dims)

    def reset(self):
        self.perf_data = PerformanceData()
        if self._session is not None:
            self._session.reset()

    # ---- tempo-token tokenizers (SPMuple2, generators.py:82-84,172-176,306-310): recognised by what they carry -------------------
    def _tracks_tempo(self):
        return hasattr(self.tokenizer, "tempos") and "Tempo" in getattr(self.tokenizer, "vocab_types_idx", {})

    def prepare_performance_notes(self, perf_idx: int, score_embeddings: Optional[Tensor] = None,
                                  perf_embeddings: Optional[Tensor] = None, overlay_bars: float = 0.5):
        perf_seq = self.dataset.performances[perf_idx]
        self.perf_data.perf_seq = perf_seq
        initial_tempo = DEFAULT_TEMPO
        if self._tracks_tempo() and hasattr(self.dataset, "initial_tempos"):
            initial_tempo = self.dataset.initial_tempos[self.dataset.performance_names[perf_idx]]
        perf_seq = self.dataset.processor.add_sos_token(perf_seq)
        perf_seq = self.dataset.processor.add_eos_token(perf_seq)
        need = self.model.perf_encoder is not None and perf_embeddings is None
        need = need or (self.model.score_encoder is not None and score_embeddings is None)
        if need:
            score_embeddings, perf_embeddings, _ = self.encode_embeddings(perf_idx, overlay_bars=overlay_bars)
        notes = np.array(perf_seq, dtype=np.int64)
        notes[1:-1, self._mask_dims] = self.collator.mask_token_id
        self._notes = notes                                                    # host copy: the loop's control state
        self.perf_data.notes = torch.from_numpy(notes).to(self.device)
        self.perf_data.embeddings = perf_embeddings.to(self.device) if perf_embeddings is not None else None
        self.perf_data.context = score_embeddings.to(self.device) if score_embeddings is not None else None
        if self._tracks_tempo():
            from types import SimpleNamespace
            self.perf_data.intermediates = SimpleNamespace(initial_tempo=initial_tempo, tempos=None)
        return self.perf_data

    # ---- window bookkeeping ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _bar_changes(bars: np.ndarray) -> np.ndarray:
        return np.nonzero(np.diff(bars))[0]

    def _window_start(self, seq: np.ndarray, max_context_len: int) -> int:
        """First kept row of the accepted sequence when it no longer fits the context: the start of the earliest bar from which the
        remainder fits (generators.py:135-140)."""
        n = seq.shape[0]
        if n < max_context_len - 1:
            return 0
        nb = self._bar_changes(seq[1:, 0])
        fits = np.nonzero(n - (nb + 1) < max_context_len)[0]
        return 0 if len(nb) == 0 or len(fits) == 0 else int(nb[fits[0]]) + 2

    def _chord_end(self, cur: int, group: bool) -> int:
        notes, end = self._notes, cur + 1
        if group:
            while end < len(notes) and notes[cur, 0] == notes[end, 0] and notes[cur, 1] == notes[end, 1]:
                end += 1
        return end

    def _session_for(self, filter_logits_fn, filter_kwargs, disable_caches):
        """The decode session when the engine covers this call: greedy (`top_k` with k = 1) or `top_k` sampling at temperature 1
        (what `unmask_tokens` is called with, generators.py:223-233); other filters take the module path."""
        if not self.use_engine or disable_caches or self.device.type != "cuda" or filter_logits_fn is not top_k:
            return None
        if self._session is None:
            from ..decode import RenderSession
            try:
                self._session = RenderSession(self.model.perf_decoder.model, self.engine_max_len, self._mask_dims.tolist(),
                                              mask_token_id=self.collator.mask_token_id)
            except NotImplementedError:
                self.use_engine = False
                return None
        try:
            if is_greedy(filter_logits_fn, filter_kwargs):
                self._session.configure(None)
            else:
                kw = dict(filter_kwargs or {})
                self._session.configure(dict(k=kw.get("k"), thres=kw.get("thres", 0.9), temperature=1.0, seed=self.sampling_seed))
        except NotImplementedError:
            return None
        return self._session

    def generate_performance_notes(self, start_time: float = 0., time_window: float = 0.2, time_window_overflow: float = 0.1,
                                   delta_embedding: Optional[Tensor] = None, max_context_len: int = 512,
                                   group_chord_notes: bool = True, time_messages: bool = True, sort_messages: bool = False,
                                   filter_logits_fn: Callable = top_k, filter_kwargs: Optional[Dict[str, object]] = None,
                                   disable_tqdm: bool = True, disable_caches: bool = False):
        pd = self.perf_data
        notes = self._notes
        has_perf_emb, has_score_emb = pd.embeddings is not None, pd.context is not None
        perf_embeddings = pd.embeddings.clone().detach() if has_perf_emb else None
        score_embeddings = pd.context
        if pd.gen_seq is None:
            pd.gen_seq = pd.notes[:1]
            self._gen = notes[:1].copy()                                       # host mirror of perf_data.gen_seq
        accepted = self._gen
        cur = accepted.shape[0]
        start_idx = self._window_start(accepted, max_context_len)
        window = accepted[start_idx:].copy()
        known = window.shape[0]
        first = int(window[0, 0] == self.sos_token_id)
        delta = None if delta_embedding is None else delta_embedding.to(self.device)
        session = self._session_for(filter_logits_fn, filter_kwargs, disable_caches)
        if session is not None and max_context_len + 64 > session.max_len:
            raise ValueError(f"max_context_len {max_context_len} exceeds the engine's window ({session.max_len}); raise engine_max_len")
        caches, intermediates = pd.caches, pd.intermediates
        if (session is None) != (not isinstance(caches, _EngineCache)) and caches is not None:
            caches = None                                                      # caches of the other path
        zero = self.tokenizer.zero_token
        produced = False
        all_times, all_tokens = [], []
        while not pd.reached_eos:
            end = self._chord_end(cur, group_chord_notes)
            new = notes[cur:end]
            n_new = new.shape[0]
            if self._tracks_tempo() and self.tokenizer.vocab_types_idx["Tempo"] not in self._mask_dims:
                tempo = intermediates.tempos[-1, 0] if intermediates.tempos is not None else intermediates.initial_tempo
                new[:, self.tokenizer.vocab_types_idx["Tempo"]] = find_closest(self.tokenizer.tempos, tempo) + zero
            if new[-1, 0] == self.eos_token_id:
                pd.reached_eos = True
                break
            window = np.concatenate([window, new], axis=0)
            last = window.shape[0]
            if last >= max_context_len:                                        # crop whole bars from the left (generators.py:184-201)
                nb = self._bar_changes(window[first:last, 0])
                shift = 1
                if len(nb) > 0:
                    fits = np.nonzero(last - (nb + first) < max_context_len)[0]
                    if len(fits) > 0 and nb[fits[0]] + 1 + first != last - 1:
                        shift = int(nb[fits[0]]) + 1 + first
                window = window[shift:]
                known, last, start_idx = known - shift, last - shift, start_idx + shift
                first, caches = 0, None
                if known < max_context_len / 8:
                    break
            bar_shift = window[first, 0] - zero                                # bars re-based to the window (generators.py:203-205)
            model_in = window.copy()
            model_in[first:last, 0] -= bar_shift
            doubled = model_in.copy()
            doubled[first:last, self._mask_dims] = self.collator.mask_token_id
            if has_perf_emb and delta is not None:
                perf_embeddings[cur:cur + n_new] += delta
            score_embs = score_embeddings[start_idx:cur + n_new] if has_score_emb else None
            perf_embs = perf_embeddings[start_idx:cur + n_new] if has_perf_emb else None

            if session is not None:
                if caches is not None and (caches.start_idx != start_idx or caches.length != last - 1 - n_new or caches.length == 0
                                           or session.tag is not caches):
                    caches = None
                if caches is None:
                    session.reset()
                    have = last - 1 - n_new                                      # rows every new note can take from the caches
                    if self.prefill == "modules" and have >= self.prefill_min and not session.cross:
                        session.load_caches(self._prefill_modules(model_in[:have + 1], doubled[:have + 1], score_embs, perf_embs))
                else:
                    session.truncate(caches.length)
                rows = session.decode(torch.from_numpy(model_in), torch.from_numpy(doubled), score_embs, perf_embs, n_new,
                                      batched_prefill=self.prefill == "engine")
                gen_tokens = rows.cpu().numpy()                                # the loop's one D2H copy per chord group (decode() has checked
                                                                               # the persistent launch's error word and re-run if it had to)
                caches = _EngineCache(start_idx, session.length)
                session.tag = caches
            else:
                gen_tokens, caches = self._unmask_modules(model_in, doubled, score_embs, perf_embs, caches, n_new, disable_caches,
                                                          filter_logits_fn, filter_kwargs, disable_tqdm)
            gen_tokens[:, 0] += bar_shift
            produced = True
            token_times, intermediates = self.messenger.tokens_to_messages(
                gen_tokens, note_attributes=False, note_off_events=False, intermediates=intermediates, return_intermediates=True, sort=False)
            all_times.extend(np.asarray(token_times).tolist())
            all_tokens.append(gen_tokens)
            if np.max(token_times) >= start_time + time_window + time_window_overflow:
                break
            window[-n_new:] = gen_tokens
            cur += n_new

        if not produced:
            return None, []
        keep = np.nonzero(np.asarray(all_times) <= start_time + time_window)[0]
        cut = 0 if len(keep) == 0 else int(keep[-1]) + 1
        if cut == 0:
            return None, []
        gen_tokens = np.concatenate(all_tokens, axis=0)[:cut]
        messages, pd.intermediates = self.messenger.tokens_to_messages(
            gen_tokens, intermediates=pd.intermediates, return_intermediates=True, to_times=time_messages, sort=sort_messages)
        total = accepted.shape[0]
        if has_perf_emb and delta is not None:
            pd.embeddings[total:total + cut] = perf_embeddings[total:total + cut]
        self._gen = np.concatenate([accepted, gen_tokens], axis=0)
        gen_seq = torch.from_numpy(gen_tokens).to(self.device)
        pd.gen_seq = torch.cat([pd.gen_seq, gen_seq])
        if caches is not None:                                                 # drop the rows of the notes that were not accepted
            dropped = len(all_times) - cut
            if isinstance(caches, _EngineCache):
                caches.length -= dropped
            else:
                caches = self.cut_caches(caches, right_idx=caches.token_emb.shape[1] - dropped)
        pd.caches = caches
        return gen_seq, messages

    def _prefill_modules(self, model_in, doubled, score_embs, perf_embs):
        """Caches of rows 0 .. len-2 of a window from one batched forward of the decoder modules (what `unmask_tokens` runs for its
        first masked position with `caches=None`, wrappers.py:391-393)."""
        from ..models.scoreperformer.embeddings import shared_tables
        n = model_in.shape[0]
        seq = torch.from_numpy(model_in).to(self.device)[None]
        dec = self.model.perf_decoder
        was_training = dec.model.training
        dec.model.eval()
        with torch.inference_mode(), shared_tables():
            out = dec(seq, seq_masked=torch.from_numpy(doubled).to(self.device)[None],
                      mask=torch.ones(1, n, dtype=torch.bool, device=self.device), return_embeddings=True, return_caches=True, caches=None,
                      context=score_embs[:n].unsqueeze(0) if score_embs is not None else None,
                      style_embeddings=perf_embs[:n].unsqueeze(0) if perf_embs is not None else None)
        dec.model.train(was_training)
        return out.caches

    def _unmask_modules(self, model_in, doubled, score_embs, perf_embs, caches, n_new, disable_caches, filter_logits_fn, filter_kwargs,
                        disable_tqdm):
        """The reference's own call (generators.py:222-241): module forward per note with concatenated caches."""
        seq = torch.from_numpy(model_in).to(self.device)
        if caches is not None:
            if seq.shape[0] - 1 - n_new != caches.token_emb.shape[1] or caches.token_emb.shape[1] == 0 \
                    or len(caches.transformer.attention) == 0:
                caches = None
        dec = self.model.perf_decoder
        engine_flag = getattr(dec, "use_decode_engine", True)
        # this IS the module path: a first call (caches=None) through the engine would hand engine-built caches to the module forward of
        # the next call, which cannot continue from them
        dec.use_decode_engine = False
        try:
            with torch.inference_mode():
                out, caches = dec.unmask_tokens(
                    seq, torch.from_numpy(doubled).to(self.device),
                    context=score_embs.unsqueeze(0) if score_embs is not None else None,
                    style_embeddings=perf_embs.unsqueeze(0) if perf_embs is not None else None,
                    caches=caches if not disable_caches else None, return_caches=True, filter_logits_fn=filter_logits_fn,
                    filter_kwargs=filter_kwargs, disable_tqdm=disable_tqdm)
        finally:
            dec.use_decode_engine = engine_flag
        return out[-n_new:].cpu().numpy(), caches

    def predict_number_of_notes(self, start_time: float = 0., time_window: float = 0.2, max_notes: int = 32):
        pd = self.perf_data
        done = len(pd.gen_seq) - 1 if pd.gen_seq is not None else 0
        future = pd.perf_seq[done:done + max_notes]
        if len(future) == 0:
            return 0.
        if pd.intermediates is not None:                                       # generators.py:306-310 (adjusts perf_seq in place)
            ti = self.tokenizer.vocab_types_idx["Tempo"]
            tempo_token = self.tokenizer[ti, f"Tempo_{int(pd.intermediates.tempos[-1, 0])}"]
            future[:, ti] += tempo_token - pd.perf_seq[done - 1, ti]
        times = self.messenger.tokens_to_messages(future, note_attributes=False, note_off_events=False,
                                                  intermediates=pd.intermediates, sort=False)
        return (np.asarray(times) <= start_time + time_window).sum()

    @staticmethod
    def cut_caches(caches, left_idx=0, right_idx=None):                        # generators.py:426-443
        right_idx = caches.token_emb.shape[-1] if right_idx is None else right_idx
        caches.token_emb = caches.token_emb[:, left_idx:right_idx]
        caches.transformer = TransformerIntermediates(
            hiddens=[t[..., left_idx:right_idx, :] for t in caches.transformer.hiddens],
            attention=[AttentionIntermediates(keys=a.keys[..., left_idx:right_idx, :], values=a.values[..., left_idx:right_idx, :])
                       for a in caches.transformer.attention])
        return caches

    def encode_embeddings(self, perf_idx: int, compute_latents: bool = False, overlay_bars: float = 0., augmentations=None):
        """Windowed encoder pass over a whole piece (generators.py:320-424): overlapping bar windows of `dataset.max_seq_len` notes
        through `model.forward_encoders`, per-note embeddings of the non-overlapping parts concatenated."""
        ds, tok = self.dataset, self.tokenizer
        perf = ds.performance_names[perf_idx]
        score, _ = ds._performance_map[perf]
        score_idx = ds.scores._name_to_idx[score]
        score_indices = ds._score_indices[score_idx]
        if score_indices is None:
            score_indices = ds.indexer.compute_bar_indices(ds.scores[score_idx])
            ds._score_indices[score_idx] = score_indices
        start_bar = 0
        end_bar = get_end_bar(score_indices, start_bar, ds.max_seq_len, ds.max_bar)
        meta = ScorePerformanceSampleMeta(idx=None, score_idx=score_idx, perf_idx=perf_idx, start_bar=start_bar, end_bar=end_bar,
                                          augmentations=augmentations)
        sample = ds.get(meta=meta)
        bar_idx, bar0 = tok.vocab_types_idx["Bar"], tok.zero_token
        score_seq = ds.scores[score_idx]

        def edges(s):
            has_sos, has_eos = s.score[0, 0] == self.sos_token_id, s.score[-1, 0] == self.eos_token_id
            return has_sos, has_eos, int(has_sos), s.score.shape[0] - int(has_eos), s.perf.shape[0] - int(has_eos)

        has_sos, has_eos, first, last, last_perf = edges(sample)
        last_bar = sample.score[-1 - int(has_eos), bar_idx] - bar0
        total_bars = score_seq[-1, bar_idx] - bar0
        emb_start_bar = start_bar
        score_parts, perf_parts = [], []
        while last_bar <= total_bars:
            inputs = self.model.allocate_inputs(self.model.prepare_inputs(self.collator((sample,))), self.device)
            shift = inputs["score"][:, first, bar_idx] - bar0
            inputs["score"][:, first:last, bar_idx] -= shift
            inputs["perf"][:, first:last_perf, bar_idx] -= shift
            with torch.inference_mode():
                enc = self.model.forward_encoders(
                    score=inputs["score"], score_mask=inputs["score_mask"], perf=inputs["perf"], perf_mask=inputs["perf_mask"],
                    bars=inputs["bars"], beats=inputs["beats"], onsets=inputs["onsets"], deadpan_mask=inputs["deadpan_mask"],
                    compute_loss=False)
            cut = 0
            if overlay_bars:
                cut = int(np.where(sample.score[:, bar_idx] - bar0 >= emb_start_bar)[0][0]) - first
            if enc.score_embeddings is not None:
                score_parts.append(enc.score_embeddings[0, cut:])
            if enc.perf_embeddings is not None:
                perf_parts.append(enc.perf_embeddings[0, cut:])
            if has_eos:
                break
            if overlay_bars:
                start_bar = sample.score[int(sample.score.shape[0] * (1 - overlay_bars)), 0] - bar0
                emb_start_bar = end_bar + 1
            else:
                emb_start_bar = start_bar = end_bar + 1
            end_bar = get_end_bar(score_indices, start_bar, ds.max_seq_len, ds.max_bar)
            meta.start_bar, meta.end_bar = start_bar, end_bar
            sample = ds.get(meta=meta)
            has_sos, has_eos, first, last, last_perf = edges(sample)
            last_bar = sample.score[last - 1, bar_idx] - bar0
        score_embeddings = torch.cat(score_parts, dim=0) if score_parts else None
        perf_embeddings = torch.cat(perf_parts, dim=0) if perf_parts else None
        latents = None
        if perf_embeddings is not None and compute_latents:
            pad = lambda s: torch.from_numpy(np.concatenate([[s[0]], s, [s[-1]]]))[None].to(self.device)
            latents = self.model.perf_encoder.embeddings_to_latents(
                embeddings=perf_embeddings[None], bars=pad(score_seq[:, 0]), beats=pad(ds._beat_maps[score_idx]),
                onsets=pad(ds._onset_maps[score_idx]))
        return score_embeddings, perf_embeddings, latents
