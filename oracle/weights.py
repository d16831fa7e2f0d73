"""Deterministic weight fill shared by the golden generator and the tests (test infrastructure).

Values are a pure function of (canonical key name, shape, seed), so a fixture never
has to store weights: the reference model (in ``make_golden.py``), the oracle and
the HIP-backed product model are all loaded from ``filled_state_dict``.
"""
import re
import zlib

import torch

_TIED = [
    (re.compile(r"^(score_encoder|perf_encoder|perf_decoder\.model|perf_decoder)\.token_emb\.embs\."), "EMBS."),
    (re.compile(r"^(perf_decoder\.model|perf_decoder)\.lm_head\.embs\."), "EMBS."),
    (re.compile(r"^(perf_decoder\.model|perf_decoder)\.lm_head\.project_emb\."), "DEC.token_emb.project_emb."),
    (re.compile(r"^(perf_decoder\.model|perf_decoder)\."), "DEC."),
    # decoder-only `Performer` (model.py:62-122): its single TupleTransformer lives under `transformer.` (wrapped: `transformer.model.`)
    (re.compile(r"^(transformer\.model|transformer)\.(token_emb|lm_head)\.embs\."), "PERFORMER.EMBS."),
    (re.compile(r"^(transformer\.model|transformer)\.lm_head\.project_emb\."), "PERFORMER.token_emb.project_emb."),
    (re.compile(r"^(transformer\.model|transformer)\."), "PERFORMER."),
]


def canonical(name: str) -> str:
    """Tied tensors (`model.py:213-218`, `embeddings.py:335-339`) map to one canonical name."""
    for pat, rep in _TIED:
        if pat.search(name):
            return pat.sub(rep, name, count=1)
    return name


def fill_like(name: str, ref: torch.Tensor, seed: int = 0) -> torch.Tensor:
    if not ref.is_floating_point() or name.endswith("token_values"):
        return ref.clone()
    g = torch.Generator().manual_seed((zlib.crc32(canonical(name).encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    r = torch.randn(ref.shape, generator=g, dtype=torch.float32)
    if name.endswith("learned_logslopes"):
        return ref.clone().float() + 0.1 * r
    if ref.ndim == 1:
        # LayerNorm weight/bias and the AdaLN bias have a deterministic 0/1 init pattern: keep it + noise;
        # nn.Linear biases are randomly initialised -> replaced entirely (so the fill is independent of RNG state)
        pattern = bool(((ref == 0) | (ref == 1)).all())
        return (ref.clone().float() if pattern else 0.) + 0.05 * r
    if name.endswith("index_weight"):
        return 0.5 * r
    fan_in = ref.shape[-1] if ref.ndim >= 2 else 1
    return r * (fan_in ** -0.5)


def filled_state_dict(model: torch.nn.Module, seed: int = 0):
    """New state_dict for `model` (call on a freshly constructed model: 1-D tensors keep their init + noise)."""
    return {k: fill_like(k, v.detach().cpu(), seed) for k, v in model.state_dict().items()}
