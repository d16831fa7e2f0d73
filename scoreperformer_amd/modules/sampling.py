"""Logit filters and sampling with the reference's names, arguments and results (`modules/sampling.py:15-59`).

They act on a [rows, V] tensor of logits (V <= 260 per key) once per generated note: host-side glue of the module decode path,
torch tensor ops on the device.  `top_k` decoding -- greedy (k = 1) or sampling -- is recognised by the render loop and served
inside the decode engine's LM-head kernel instead (`spn_dec_head` / `spn_dec_head_sample`, see decode.py).
"""
import math
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

NEG_INF = float("-inf")


def top_p(logits: Tensor, thres: float = 0.9):
    """Nucleus filter: the smallest set of largest logits whose probability mass exceeds `thres`; the rest become -inf."""
    order = torch.argsort(logits, dim=-1, descending=True)
    ranked = torch.gather(logits, -1, order)
    mass_before = torch.cumsum(torch.softmax(ranked, dim=-1), dim=-1) - torch.softmax(ranked, dim=-1)
    ranked = ranked.masked_fill(mass_before > thres, NEG_INF)      # an entry is dropped when the mass ahead of it already exceeds thres
    return torch.empty_like(logits).scatter_(-1, order, ranked)


def top_k(logits: Tensor, thres: float = 0.9, k: Optional[int] = None):
    """Keep the k largest logits per row (k given, or the top (1 - thres) share of the vocabulary), -inf elsewhere."""
    if k is None:
        k = math.ceil((1 - thres) * logits.shape[-1])
    best = torch.topk(logits, k, dim=-1)
    return torch.full_like(logits, NEG_INF).scatter(-1, best.indices, best.values)


def top_a(logits: Tensor, min_p_pow: float = 2.0, min_p_ratio: float = 0.02):
    """Drop entries whose probability is below min_p_ratio * (largest probability) ** min_p_pow."""
    probs = torch.softmax(logits, dim=-1)
    floor = probs.max().pow(min_p_pow) * min_p_ratio
    return logits.masked_fill(probs < floor, NEG_INF)


def is_greedy(filter_logits_fn: Callable, filter_kwargs: Optional[Dict[str, object]]) -> bool:
    """`top_k` with k = 1: the multinomial draw over a one-hot distribution is the arg-max."""
    return filter_logits_fn is top_k and (filter_kwargs or {}).get("k") == 1


def filter_logits_and_sample(logits: Tensor, filter_logits_fn: Callable, filter_kwargs: Optional[Dict[str, object]] = None,
                             temperature: float = 1., sample: bool = True):
    """Filter, temper, normalise; one multinomial draw per row, or the probabilities themselves with sample=False."""
    kept = filter_logits_fn(logits, **(filter_kwargs or {}))
    probs = torch.softmax(kept / temperature, dim=-1)
    return torch.multinomial(probs, 1) if sample else probs
