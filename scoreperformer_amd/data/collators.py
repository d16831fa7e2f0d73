"""Device-side mirror of the reference's MixedLM score-performance collator.

Same class name, constructor arguments, call signature and result fields as
``scoreperformer/data/collators/score_performance.py:186-234`` (``MixedLMScorePerformanceCollator``) and its bases
(``performance.py:18-92, 213-247``), but the padded batch is built in HBM: the host only concatenates the ragged samples into ONE
pinned int32 staging buffer (tokens, segment ids, row offsets, deadpan flags), one H2D copy moves it, and
``spn_collate_mixlm`` (csrc/collate.hip) writes every tensor of the batch.  The result is already on the device, so
``model.allocate_inputs`` is a no-op for it.  No CPU fallback: without the HIP library this raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from .. import ops
from ..lib import SpnError


@dataclass
class SeqInputs:                                   # data/collators/common.py:8-12
    tokens: torch.Tensor
    mask: torch.Tensor
    lengths: torch.Tensor


class SegmentBounds(dict):
    """{"bar" | "beat" | "onset": largest segment id of the batch + 1} as python ints, taken from the host-side segment arrays the
    collator concatenates anyway.  The reference reads `segments.max() + 1` back from the device in every forward
    (models/scoreperformer/mmd_transformer.py:330); with these the forward needs no host read.  `.to()` returns the object itself so
    that the trainer's `allocate_inputs` (`value.to(device)` over the input dict, models/base.py:26-27) passes it through."""

    def to(self, *args, **kwargs):
        return self


@dataclass
class SeqSegments:                                 # score_performance.py:18-22
    bar: Optional[torch.Tensor] = None
    beat: Optional[torch.Tensor] = None
    onset: Optional[torch.Tensor] = None
    bounds: Optional[SegmentBounds] = None         # not in the reference: host-known slot counts (see SegmentBounds)


@dataclass
class MixedLMScorePerformanceInputs:               # score_performance.py:25-32,118-120,180-182 (flattened hierarchy)
    scores: SeqInputs
    performances: SeqInputs
    noisy_performances: Optional[SeqInputs] = None
    segments: Optional[SeqSegments] = None
    directions: Optional[Union[Dict[str, torch.Tensor], torch.Tensor]] = None
    deadpan_mask: Optional[torch.Tensor] = None
    labels: Optional[SeqInputs] = None
    masked_performances: Optional[SeqInputs] = None


class MixedLMScorePerformanceCollator:
    def __init__(
            self,
            pad_token_id: int = 0,
            pad_to_multiple_of: int = 1,

            mask_token_id: int = 1,
            mask_ignore_token_ids: Optional[List[int]] = None,
            mask_ignore_token_dims: Optional[List[int]] = None,
            label_pad_ignored_dims: bool = True,
            label_pad_token_id: int = -100,

            device: Union[str, torch.device] = "cuda"
    ):
        self.pad_token_id = pad_token_id
        self.pad_to_multiple_of = pad_to_multiple_of
        self.mask_token_id = mask_token_id
        self.mask_ignore_token_ids = {*(mask_ignore_token_ids or []), pad_token_id}       # performance.py:230
        self.mask_ignore_token_dims = mask_ignore_token_dims or []
        self.label_pad_ignored_dims = label_pad_ignored_dims
        self.label_pad_token_id = label_pad_token_id
        self.device = torch.device(device)
        self._staging: Optional[torch.Tensor] = None                                     # pinned int32, grown on demand
        self._staged = None                                                              # event: last H2D copy of the staging buffer

    def pad_len(self, length):                      # performance.py:28-33
        if self.pad_to_multiple_of > 0:
            pad_size = self.pad_to_multiple_of - length % self.pad_to_multiple_of
            length += pad_size if 0 < pad_size < self.pad_to_multiple_of else 0
        return int(length)

    def get_max_lengths(self, batch, inference=False):  # performance.py:35-40, score_performance.py:44-54
        max_perf = max(len(s.perf) for s in batch)
        return {"performance": int(max_perf) if inference else self.pad_len(max_perf),
                "score": self.pad_len(max(len(s.score) for s in batch))}

    def _stage(self, words: int) -> torch.Tensor:
        if self._staged is not None:
            self._staged.synchronize()              # the previous batch's H2D copy must have read the buffer
        if self._staging is None or self._staging.numel() < words:
            self._staging = torch.empty(max(words, 1 << 16), dtype=torch.int32).pin_memory()
        return self._staging[:words]

    def __call__(self, batch, inference: bool = False, return_tensors: bool = True) -> MixedLMScorePerformanceInputs:
        if not torch.cuda.is_available():
            raise SpnError("MixedLMScorePerformanceCollator builds the batch on the GPU; no HIP device is visible (no CPU fallback)")
        if any(getattr(s, "directions", None) is not None for s in batch):
            raise NotImplementedError("score directions (classifier labels) are outside the accelerated path (SURVEY.md section 8)")
        has_noisy = all(getattr(s, "noisy_perf", None) is not None for s in batch)      # score_performance.py:48
        b = len(batch)
        Ks, Kp = batch[0].score.shape[-1], batch[0].perf.shape[-1]
        if batch[0].score.ndim != 2 or batch[0].perf.ndim != 2:
            raise NotImplementedError("the device collator takes tuple tokens [n, K] (OctupleM)")
        has_seg = batch[0].segments is not None
        lens = self.get_max_lengths(batch, inference=inference)
        n_s = np.fromiter((len(s.score) for s in batch), np.int64, b)
        n_p = np.fromiter((len(s.perf) for s in batch), np.int64, b)
        sum_s, sum_p = int(n_s.sum()), int(n_p.sum())

        # one pinned staging buffer: score tokens | perf tokens | bar,beat,onset | score offsets | perf offsets | deadpan bytes
        o_perf = sum_s * Ks
        o_seg = o_perf + sum_p * Kp
        o_soff = o_seg + (3 * sum_s if has_seg else 0)
        o_poff = o_soff + b + 1
        o_dead = o_poff + b + 1
        o_noisy = o_dead + (b + 3) // 4
        n_n = np.fromiter((len(s.noisy_perf) for s in batch), np.int64, b) if has_noisy else None
        sum_n = int(n_n.sum()) if has_noisy else 0
        o_noff = o_noisy + sum_n * Kp
        words = o_noff + (b + 1 if has_noisy else 0)
        host = self._stage(words)
        h = host.numpy()
        np.concatenate([s.score for s in batch], out=h[:o_perf].reshape(sum_s, Ks), casting="unsafe")
        np.concatenate([s.perf for s in batch], out=h[o_perf:o_seg].reshape(sum_p, Kp), casting="unsafe")
        bounds = None
        if has_seg:
            bounds = SegmentBounds()
            for j, name in enumerate(("bar", "beat", "onset")):
                col = h[o_seg + j * sum_s:o_seg + (j + 1) * sum_s]
                np.concatenate([getattr(s.segments, name) for s in batch], out=col, casting="unsafe")
                bounds[name] = (max(int(col.max()), 0) if sum_s else 0) + 1        # padded positions hold segment id 0
        h[o_soff] = 0
        np.cumsum(n_s, out=h[o_soff + 1:o_poff], dtype=np.int32)
        h[o_poff] = 0
        np.cumsum(n_p, out=h[o_poff + 1:o_dead], dtype=np.int32)
        dead = h[o_dead:o_noisy].view(np.uint8)
        dead[:b] = [bool(s.is_deadpan) for s in batch]
        if has_noisy:
            np.concatenate([s.noisy_perf for s in batch], out=h[o_noisy:o_noff].reshape(sum_n, Kp), casting="unsafe")
            h[o_noff] = 0
            np.cumsum(n_n, out=h[o_noff + 1:words], dtype=np.int32)

        dev = host.to(self.device, non_blocking=True)
        self._staged = torch.cuda.Event()
        self._staged.record()
        dims = 0
        for d in self.mask_ignore_token_dims:
            dims |= 1 << (d % Kp)
        t = ops.collate_mixlm(
            dev[:o_perf], dev[o_perf:o_seg], dev[o_seg:o_soff] if has_seg else None, dev[o_soff:o_poff], dev[o_poff:o_dead],
            dev[o_dead:o_noisy].view(torch.uint8), b=b, Ks=Ks, Kp=Kp, Ls=lens["score"], Lp=lens["performance"], pad_id=self.pad_token_id,
            mask_id=self.mask_token_id, label_pad_id=self.label_pad_token_id,
            ignore_ids=sorted(self.mask_ignore_token_ids - {self.pad_token_id}), ignore_dims=dims,
            label_pad_ignored_dims=self.label_pad_ignored_dims)
        perf = SeqInputs(t["perf"], t["perf_mask"], t["perf_len"])
        noisy = None
        if has_noisy:   # padded like the score (always to the multiple, score_performance.py:50)
            noisy = SeqInputs(*ops.collate_pad_tokens(dev[o_noisy:o_noff], dev[o_noff:words], b=b, K=Kp, L=self.pad_len(int(n_n.max())),
                                                      pad_id=self.pad_token_id))
        return MixedLMScorePerformanceInputs(
            scores=SeqInputs(t["score"], t["score_mask"], t["score_len"]),
            performances=perf,
            noisy_performances=noisy,
            segments=SeqSegments(t["bar"], t["beat"], t["onset"], bounds) if has_seg else None,
            deadpan_mask=t["deadpan_mask"],
            masked_performances=SeqInputs(t["masked_perf"], t["perf_mask"].clone(), t["perf_len"]),   # score_performance.py:212
            labels=SeqInputs(t["labels"], perf.mask, perf.lengths),
        )
