"""Golden vectors of the reference's windowed encoder pass, ScorePerformerGenerator.encode_embeddings (authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_encode

Real reference generator + encoders (CPU fp32) over `oracle/render_fakes.FakeScoreDataset`; records the concatenated per-note score
and performance embeddings for overlay_bars = 0 and 0.5 (tests/golden/encode_embeddings.npz).  Data only.
"""
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
from oracle.refimport import stubs  # noqa: E402

stubs.install()
warnings.filterwarnings("ignore")
from scoreperformer.data.collators.score_performance import MixedLMScorePerformanceCollator  # noqa: E402
from scoreperformer.inference.generators import ScorePerformerGenerator  # noqa: E402

from oracle.refimport.make_golden import SMALL_VOCAB, build  # noqa: E402
from oracle.refimport.make_golden_render import COLLATOR  # noqa: E402
from oracle.render_fakes import FakeMessenger, FakeScoreDataset, make_piece  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")


def main():
    cfg, model, kw = build(dict(preset="tiny"), seed=3)
    model.eval()
    piece = make_piece(51, 130, SMALL_VOCAB)
    out = {"piece": piece, "max_seq_len": np.array(48)}
    for overlay in (0.0, 0.5):
        ds = FakeScoreDataset(SMALL_VOCAB, piece, max_seq_len=48)
        gen = ScorePerformerGenerator(model, ds, MixedLMScorePerformanceCollator(**COLLATOR), FakeMessenger(SMALL_VOCAB), device="cpu")
        with torch.no_grad():
            se, pe, lat = gen.encode_embeddings(0, compute_latents=True, overlay_bars=overlay)
        out[f"overlay{overlay}/score_emb"], out[f"overlay{overlay}/perf_emb"] = se.numpy(), pe.numpy()
        for i, z in enumerate(lat if isinstance(lat, (list, tuple)) else [lat]):
            out[f"overlay{overlay}/latent{i}"] = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        print("overlay", overlay, "score_emb", tuple(se.shape), "perf_emb", tuple(pe.shape), "latents", type(lat).__name__)
    path = os.path.join(OUT, "encode_embeddings.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
