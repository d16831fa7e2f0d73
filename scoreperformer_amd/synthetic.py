"""Synthetic configs and batches for the ScorePerformer hot path.

* ``model_config(preset)`` builds the dict config that mirrors
  `recipes/scoreperformer/base.yaml:68-192` with interpolations resolved by hand
  (SURVEY.md Appendix B), for the BASELINE.json configurations C1..C5.
* ``synthetic_batch`` produces the input contract of
  `ScorePerformer.prepare_inputs` (`models/scoreperformer/model.py:343-372`) the way
  `MixedLMScorePerformanceCollator` would (`data/collators/score_performance.py:209-234`,
  `data/collators/performance.py:239-255`): uniform random tokens, MASKed
  performance dims {3,5,10,11}, -100 labels elsewhere, monotone segment ids.
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch

from .utils.config import OmegaConf

# SPMuple(Window) vocabulary sizes incl. 4 specials (derived from
# data/tokenizers/spmuple_window.json + octuple_m.py:295-345, spmuple.py:591-651).
PERFORMANCE_VOCAB: Dict[str, int] = {
    "Bar": 260, "Position": 132, "Pitch": 92, "Velocity": 132, "Duration": 133, "Tempo": 125,
    "TimeSig": 26, "PositionShift": 69, "NotesInOnset": 16, "PositionInOnset": 16,
    "RelOnsetDev": 165, "RelPerfDuration": 85,
}
SCORE_KEYS = list(PERFORMANCE_VOCAB)[:10]
PAD, MASK, SOS, EOS = 0, 1, 2, 3
# `mask_ignore_token_dims` of recipes/scoreperformer/base.yaml:65 -> predicted dims
PREDICTED_DIMS = (3, 5, 10, 11)
IGNORED_DIMS = (0, 1, 2, 4, 6, 7, 8, 9)

PRESETS = {
    # name: dim, heads, (score, perf_enc, dec) depth, emb_dims, latent dims, max_seq_len
    "tiny": dict(dim=128, heads=2, depths=(2, 2, 2), emb_dims=32, latent_dim=[16, 8, 4, 4], max_seq_len=256),
    "base": dict(dim=256, heads=4, depths=(2, 4, 4), emb_dims=128, latent_dim=[32, 20, 8, 4], max_seq_len=256),
    "c2": dict(dim=512, heads=8, depths=(6, 6, 6), emb_dims=128, latent_dim=[32, 20, 8, 4], max_seq_len=1024),
    "c3": dict(dim=512, heads=8, depths=(6, 6, 6), emb_dims=128, latent_dim=[32, 20, 8, 4], max_seq_len=2048),
    "c5": dict(dim=512, heads=8, depths=(6, 6, 6), emb_dims=128, latent_dim=[32, 20, 8, 4], max_seq_len=4096),
}


def model_config(
        preset: str = "tiny",
        *,
        context_emb_mode: str = "cat",
        style_emb_mode: str = "adanorm",
        dropout: float = 0.0,
        latent_dropout: Optional[list] = None,
        one_kv_head: bool = True,
        alibi_learned: bool = True,
        lm_head: str = "lm-tied",
        mode: str = "mixlm",
        num_tokens: Optional[Dict[str, int]] = None,
        **overrides,
):
    """Dict config for ``ScorePerformer.init`` (structure of base.yaml:68-192)."""
    p = dict(PRESETS[preset]); p.update(overrides)
    num_tokens = dict(num_tokens or PERFORMANCE_VOCAB)
    score_tokens = {k: num_tokens[k] for k in list(num_tokens)[:10]}
    te = dict(_target_="simple", emb_dims=p["emb_dims"], mode="cat", emb_norm=True, discrete=False,
              continuous=True, continuous_dense=True, discrete_ids=[0, 1, 2, 3], tie_keys=None,
              token_values=None)
    attn = dict(dim_head=64, one_kv_head=one_kv_head, dropout=dropout, alibi_pos_bias=True,
                alibi_learned=alibi_learned)
    ff = dict(mult=4, glu=True, swish=True, dropout=dropout)

    def tr(target, depth):
        return dict(_target_=target, depth=depth, heads=p["heads"], attention=copy.deepcopy(attn),
                    feed_forward=copy.deepcopy(ff))

    n_lat = len(p["latent_dim"])
    cfg = dict(
        dim=p["dim"], tie_token_emb=True, mode=mode, num_tokens=num_tokens, num_score_tokens=score_tokens,
        classifiers=None,
        score_encoder=dict(token_embeddings=copy.deepcopy(te), emb_norm=True, emb_dropout=0, use_abs_pos_emb=False,
                           max_seq_len=p["max_seq_len"], transformer=tr("encoder", p["depths"][0])),
        perf_encoder=dict(token_embeddings=copy.deepcopy(te), emb_norm=True, emb_dropout=0, use_abs_pos_emb=False,
                          max_seq_len=p["max_seq_len"],
                          latent_dim=list(p["latent_dim"]),
                          aggregate_mode=["mean", "bar_mean", "beat_mean", "onset_mean"][:n_lat],
                          latent_dropout=list(latent_dropout) if latent_dropout is not None else [0.0] * n_lat,
                          hierarchical=True, inclusive_latent_dropout=True, deadpan_zero_latent=True, loss_weight=1.,
                          transformer=tr("encoder", p["depths"][1])),
        perf_decoder=dict(token_embeddings=dict(te, _target_="multi-seq", multiseq_mode="post-cat"),
                          emb_norm=True, emb_dropout=0, use_abs_pos_emb=False, max_seq_len=p["max_seq_len"],
                          context_emb_mode=context_emb_mode, style_emb_dim=sum(p["latent_dim"]),
                          style_emb_mode=style_emb_mode, transformer=tr("decoder", p["depths"][2]),
                          lm_head=dict(_target_=lm_head)),
    )
    return OmegaConf.create(cfg)


def synthetic_batch(
        batch: int,
        seq_len: int,
        *,
        seed: int = 1234,
        num_tokens: Optional[Dict[str, int]] = None,
        ragged: bool = False,
        sos: bool = True,
        deadpan_p: float = 0.0,
        device: str = "cpu",
        with_bounds: bool = False,
):
    """Seeded synthetic MixedLM batch (SURVEY.md §8(d) 'Concrete synthetic inputs').  `with_bounds`: also `segment_bounds`, the
    host-known slot counts the device collator emits with a real batch (`data.SegmentBounds`: the forward then reads nothing back)."""
    num_tokens = dict(num_tokens or PERFORMANCE_VOCAB)
    g = torch.Generator().manual_seed(seed)
    sizes = list(num_tokens.values())
    perf = torch.stack([torch.randint(4, v, (batch, seq_len), generator=g) for v in sizes], dim=-1)
    if sos:
        perf[:, 0] = SOS
    lengths = torch.full((batch,), seq_len)
    if ragged:
        lengths = torch.randint(seq_len // 2, seq_len + 1, (batch,), generator=g)
    mask = torch.arange(seq_len)[None, :] < lengths[:, None]
    perf = perf * mask[..., None]

    def segs(p):
        inc = (torch.rand(batch, seq_len, generator=g) < p).long()
        inc[:, 0] = 0
        return (4 + torch.cumsum(inc, dim=1)) * mask

    bars, beats, onsets = segs(1 / 16), segs(1 / 4), segs(1 / 2)
    score = perf[..., :10].clone()

    # element-wise special-token test, as `mask_with_tokens(..., squeeze=False)` does
    special = perf <= EOS
    masked_perf = perf.clone()
    labels = perf.clone()
    for d in PREDICTED_DIMS:
        masked_perf[..., d] = torch.where(special[..., d], perf[..., d], torch.full_like(perf[..., d], MASK))
    for d in IGNORED_DIMS:
        labels[..., d] = -100
    labels[special] = -100

    deadpan = torch.rand(batch, generator=g) < deadpan_p
    out = dict(perf=perf, perf_mask=mask, score=score, score_mask=mask.clone(), masked_perf=masked_perf,
               labels=labels, bars=bars, beats=beats, onsets=onsets, deadpan_mask=deadpan)
    out = {k: v.to(device) for k, v in out.items()}
    if with_bounds:
        from .data.collators import SegmentBounds
        out["segment_bounds"] = SegmentBounds(bar=int(bars.max()) + 1, beat=int(beats.max()) + 1, onset=int(onsets.max()) + 1)
    return out
