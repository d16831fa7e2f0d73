"""`Performer` / `ScorePerformer` with the reference's contract (`models/scoreperformer/model.py:50-407`)."""
from dataclasses import dataclass
from typing import Optional, Dict, Union

import torch
from torch import Tensor

from ...modules.constructor import ModuleConfig
from ...utils import default
from ...utils.config import DictConfig, MISSING
from ..base import Model
from .embeddings import TupleTokenLMHeadConfig, shared_tables
from .mmd_transformer import MMDTupleTransformer, MMDTupleTransformerOutput
from .transformer import TupleTransformerConfig, TupleTransformerOutput, TupleTransformer
from .wrappers import LMWrapper, ScorePerformerLMModes, ScorePerformerLMWrappers, finalize_lm_losses, mark_inputs_ready
from ...utils.amp import no_autocast


def _get(inputs, *path):
    cur = inputs
    for p in path:
        cur = cur[p] if isinstance(cur, dict) else getattr(cur, p)
    return cur


@dataclass
class PerformerConfig(ModuleConfig):
    transformer: TupleTransformerConfig = MISSING
    mode: Optional[str] = None


@dataclass
class PerformerOutputs(TupleTransformerOutput):
    loss: Optional[Tensor] = None
    losses: Optional[Dict[str, Tensor]] = None


class _LMModeMixin:
    _lm_attr = "perf_decoder"

    def _prepare_for_lm(self, mode):
        cur = getattr(self, self._lm_attr)
        if isinstance(cur, LMWrapper):
            cur = cur.model
        setattr(self, self._lm_attr, ScorePerformerLMWrappers[ScorePerformerLMModes(mode)](cur))
        self.mode = mode

    def prepare_for_mlm(self):
        self._prepare_for_lm(ScorePerformerLMModes.MLM)

    def prepare_for_clm(self):
        self._prepare_for_lm(ScorePerformerLMModes.CLM)

    def prepare_for_mixlm(self):
        self._prepare_for_lm(ScorePerformerLMModes.MixedLM)

    def _init_mode(self, mode):
        self.mode = mode
        if mode == ScorePerformerLMModes.MLM:
            self.prepare_for_mlm()
        elif mode == ScorePerformerLMModes.CLM:
            self.prepare_for_clm()
        elif mode == ScorePerformerLMModes.MixedLM:
            self.prepare_for_mixlm()


class Performer(_LMModeMixin, Model):
    _lm_attr = "transformer"

    def __init__(self, transformer: Union[DictConfig, TupleTransformerConfig], mode: Optional[str] = None):
        super().__init__()
        self.transformer = TupleTransformer.init(
            transformer, lm_head=transformer.get("lm_head", TupleTokenLMHeadConfig(dim=transformer.dim)))
        self._init_mode(mode)

    @no_autocast
    def forward(self, perf: Tensor, mask: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                masked_perf: Optional[Tensor] = None):
        with shared_tables():
            if masked_perf is not None:
                return self.transformer(perf, mask=mask, labels=labels, seq_masked=masked_perf)
            return self.transformer(perf, mask=mask, labels=labels)

    def prepare_inputs(self, inputs):
        d = {"perf": _get(inputs, "performances", "tokens"), "mask": _get(inputs, "performances", "mask")}
        if hasattr(inputs, "labels"):
            d["labels"] = inputs.labels.tokens
        if hasattr(inputs, "masked_performances"):
            d["masked_perf"] = inputs.masked_performances.tokens
        return d


@dataclass
class ScorePerformerConfig(ModuleConfig):
    num_tokens: Dict[str, int] = MISSING
    dim: int = MISSING
    perf_decoder: TupleTransformerConfig = MISSING
    score_encoder: Optional[TupleTransformerConfig] = None
    perf_encoder: Optional[TupleTransformerConfig] = None
    classifiers: Optional[DictConfig] = None
    tie_token_emb: bool = False
    mode: Optional[str] = None
    num_score_tokens: Optional[Dict[str, int]] = None


@dataclass
class ScorePerformerEncoderOutputs:
    score_embeddings: Optional[Tensor] = None
    score_mask: Optional[Tensor] = None
    perf_embeddings: Optional[Tensor] = None
    score_encoder: Optional[TupleTransformerOutput] = None
    perf_encoder: Optional[MMDTupleTransformerOutput] = None


@dataclass
class ScorePerformerOutputs:
    perf_decoder: TupleTransformerOutput
    score_encoder: Optional[TupleTransformerOutput] = None
    perf_encoder: Optional[MMDTupleTransformerOutput] = None
    classifiers: Optional[object] = None
    loss: Optional[Tensor] = None
    losses: Optional[Dict[str, Tensor]] = None


class ScorePerformer(_LMModeMixin, Model):
    def __init__(self, num_tokens: Dict[str, int], dim: int, perf_decoder, score_encoder=None, perf_encoder=None,
                 classifiers=None, tie_token_emb: bool = False, mode: Optional[str] = None,
                 num_score_tokens: Optional[Dict[str, int]] = None):
        super().__init__()
        self.score_encoder = None
        if score_encoder is not None:
            self.score_encoder = TupleTransformer.init(score_encoder, num_tokens=num_score_tokens or num_tokens, dim=dim,
                                                       lm_head=None)
        self.perf_encoder = None
        if perf_encoder is not None:
            self.perf_encoder = MMDTupleTransformer.init(perf_encoder, num_tokens=num_tokens, dim=dim, lm_head=None)
        self.classifiers = None
        if classifiers is not None and classifiers.get("num_classes", None) is not None:
            # direction classifiers need labels parsed from real MusicXML scores (out of scope)
            raise NotImplementedError("direction classifiers are outside the hot path of this build")
        perf_decoder.transformer.cross_attend = self.score_encoder is not None
        context_emb_dim = None if self.score_encoder is None else self.score_encoder.dim
        style_emb_dim = None if self.perf_encoder is None else self.perf_encoder.embedding_dim
        self.perf_decoder = TupleTransformer.init(
            perf_decoder, num_tokens=num_tokens, dim=dim, context_emb_dim=context_emb_dim, style_emb_dim=style_emb_dim,
            lm_head=perf_decoder.get("lm_head", TupleTokenLMHeadConfig(dim=dim)))
        if tie_token_emb:
            for key, emb in self.perf_decoder.token_emb.embs.items():
                if self.score_encoder is not None and key in self.score_encoder.token_emb.embs:
                    self.score_encoder.token_emb.embs[key] = emb
                if self.perf_encoder is not None and key in self.perf_encoder.token_emb.embs:
                    self.perf_encoder.token_emb.embs[key] = emb
        self.sync_free = False  # True: keep every loss key, never read flags back to the host inside forward
        self._init_mode(mode)

    @no_autocast
    def forward_encoders(self, perf=None, perf_mask=None, score=None, score_mask=None, bars=None, beats=None, onsets=None,
                         deadpan_mask=None, compute_loss: bool = True, segment_bounds=None):
        score_emb = perf_emb = None
        score_enc_out = perf_enc_out = None
        with shared_tables():
            if self.score_encoder is not None:
                score_enc_out = self.score_encoder(score, mask=score_mask)
                score_emb = score_enc_out.hidden_state
            if self.perf_encoder is not None:
                extra = {} if segment_bounds is None else {"segment_bounds": segment_bounds}
                perf_enc_out = self.perf_encoder(perf, mask=perf_mask, bars=bars, beats=beats, onsets=onsets,
                                                 deadpan_mask=deadpan_mask, compute_loss=compute_loss, **extra)
                perf_emb = perf_enc_out.embeddings
        return ScorePerformerEncoderOutputs(score_embeddings=score_emb, score_mask=score_mask, perf_embeddings=perf_emb,
                                            score_encoder=score_enc_out, perf_encoder=perf_enc_out)

    @no_autocast
    def forward(self, perf: Tensor, perf_mask=None, score=None, score_mask=None, noisy_perf=None, noisy_perf_mask=None,
                masked_perf=None, labels=None, bars=None, beats=None, onsets=None, directions=None, deadpan_mask=None,
                segment_bounds=None):
        """The reference's signature (model.py:280-293) plus `segment_bounds`: {"bar" | "beat" | "onset": max id + 1} python ints from
        the input pipeline (`data.SegmentBounds`, emitted by the collator) -- with them the style encoder sizes its segment slots without
        the reference's device read-back (mmd_transformer.py:330)."""
        if labels is not None and labels.is_cuda:
            mark_inputs_ready()
        with shared_tables():
            enc_out = self.forward_encoders(
                perf=default(noisy_perf, perf), perf_mask=default(noisy_perf_mask, perf_mask), score=score,
                score_mask=score_mask, bars=bars, beats=beats, onsets=onsets, deadpan_mask=deadpan_mask,
                segment_bounds=segment_bounds)
            dec_kwargs = dict(mask=perf_mask, style_embeddings=enc_out.perf_embeddings, context=enc_out.score_embeddings,
                              context_mask=enc_out.score_mask, labels=labels)
            if masked_perf is not None:
                dec_kwargs["seq_masked"] = masked_perf
            perf_dec_out = self.perf_decoder(perf, _defer_sync=True, **dec_kwargs)
        loss, losses = perf_dec_out.loss, perf_dec_out.losses
        pe = enc_out.perf_encoder
        if pe is not None and pe.loss is not None:
            loss = loss + pe.loss
            losses.update(**pe.losses)
        # ONE host read for the whole forward: which CE keys had labels, which deadpan terms are non-zero
        if not self.sync_free and labels is not None:
            ce_keys = list(perf_dec_out._ce_keys)
            flags = getattr(pe, "_flags", {}) if pe is not None else {}
            host = torch.stack([perf_dec_out.ce_sums[k][1] for k in ce_keys]
                               + [f.float() for f in flags.values()]).tolist()
            finalize_lm_losses(perf_dec_out, host[:len(ce_keys)])
            dead = {k for k, v in zip(flags.keys(), host[len(ce_keys):]) if v == 0}
            losses = {k: v for k, v in losses.items() if k not in dead and (k in perf_dec_out.losses or k not in ce_keys)}
            if pe is not None and pe.losses is not None:
                for k in dead:
                    pe.losses.pop(k, None)
        perf_dec_out.loss = loss
        return ScorePerformerOutputs(perf_decoder=perf_dec_out, score_encoder=enc_out.score_encoder, perf_encoder=pe,
                                     classifiers=None, loss=loss, losses=losses)

    def prepare_inputs(self, inputs):
        if isinstance(inputs, dict):
            return inputs
        d = {"perf": inputs.performances.tokens, "perf_mask": inputs.performances.mask,
             "score": inputs.scores.tokens, "score_mask": inputs.scores.mask}
        if getattr(inputs, "labels", None) is not None:
            d["labels"] = inputs.labels.tokens
        if getattr(inputs, "noisy_performances", None) is not None:
            d["noisy_perf"] = inputs.noisy_performances.tokens
            d["noisy_perf_mask"] = inputs.noisy_performances.mask
        if getattr(inputs, "masked_performances", None) is not None:
            d["masked_perf"] = inputs.masked_performances.tokens
        if getattr(inputs, "segments", None) is not None:
            d["bars"], d["beats"], d["onsets"] = inputs.segments.bar, inputs.segments.beat, inputs.segments.onset
            if getattr(inputs.segments, "bounds", None) is not None:   # data.SegmentBounds of the device collator: no read-back in forward
                d["segment_bounds"] = inputs.segments.bounds
        if getattr(inputs, "directions", None) is not None:
            d["directions"] = inputs.directions
        if getattr(inputs, "deadpan_mask", None) is not None:
            d["deadpan_mask"] = inputs.deadpan_mask
        return d

    @staticmethod
    def inject_data_config(config, dataset):
        config["num_tokens"] = dataset.tokenizer.performance_sizes
        config["num_score_tokens"] = dataset.tokenizer.score_sizes
        for key in ["score_encoder", "perf_encoder", "perf_decoder"]:
            if config.get(key) is not None:
                config[key]["token_embeddings"]["token_values"] = {
                    k: v.tolist() for k, v in dataset.tokenizer.token_values(normalize=True).items()}
        return config

    @staticmethod
    def cleanup_config(config):
        for key in ["score_encoder", "perf_encoder", "perf_decoder"]:
            if config.get(key) is not None:
                del config[key]["token_embeddings"]["token_values"]
        return config
