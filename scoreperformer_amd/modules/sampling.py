"""Logit filters and sampling with the reference's signatures (`modules/sampling.py:15-59`).

These run on device tensors of one row of per-key logits (V <= 260) once per generated note; they are host-side glue
around the decode kernels, written with torch tensor ops on the GPU.  Greedy decoding (`top_k` with k=1 followed by
multinomial over a one-hot distribution) is recognised by the render loop and served by the fused argmax of the head
kernel instead (see models/scoreperformer/wrappers.py).
"""
import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from ..utils import default


def top_p(logits: Tensor, thres: float = 0.9):
    sorted_logits, sorted_indices = torch.sort(logits, descending=True)
    cum_probs = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
    remove = cum_probs > thres
    remove = F.pad(remove, (1, -1), value=False)
    sorted_logits[remove] = float("-inf")
    return sorted_logits.scatter(1, sorted_indices, sorted_logits)


def top_k(logits: Tensor, thres: float = 0.9, k: Optional[int] = None):
    k = default(k, math.ceil((1 - thres) * logits.shape[-1]))
    val, ind = torch.topk(logits, k)
    probs = torch.full_like(logits, float("-inf"))
    probs.scatter_(1, ind, val)
    return probs


def top_a(logits: Tensor, min_p_pow: float = 2.0, min_p_ratio: float = 0.02):
    probs = F.softmax(logits, dim=-1)
    limit = torch.pow(torch.max(probs), min_p_pow) * min_p_ratio
    return torch.where(probs < limit, float("-inf"), logits)


def is_greedy(filter_logits_fn: Callable, filter_kwargs: Optional[Dict[str, object]]) -> bool:
    return filter_logits_fn is top_k and (filter_kwargs or {}).get("k", None) == 1


def filter_logits_and_sample(logits: Tensor, filter_logits_fn: Callable, filter_kwargs: Optional[Dict[str, object]] = None,
                             temperature: float = 1., sample: bool = True):
    filter_kwargs = filter_kwargs or {}
    filtered = filter_logits_fn(logits, **filter_kwargs)
    probs = F.softmax(filtered / temperature, dim=-1)
    if not sample:
        return probs
    return torch.multinomial(probs, 1)
