"""Golden vectors of the reference's render loop, ScorePerformerGenerator.generate_performance_notes (authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_render

Runs the REAL reference generator (inference/generators.py:106-295) and decoder on CPU fp32, with the build-owned stand-ins of
oracle/render_fakes.py for tokenizer / messenger / dataset, over a whole piece in successive time windows, and records every call's
generated tokens and messages (tests/golden/render_loop.npz).  Data only.
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
from scoreperformer.modules.sampling import top_k  # noqa: E402

from oracle.refimport.make_golden import SMALL_VOCAB, build  # noqa: E402
from oracle.render_fakes import FakeMessenger, make_dataset, make_piece  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
COLLATOR = dict(pad_token_id=0, pad_to_multiple_of=1, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
                mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9])
SCENARIOS = {
    # name: (piece seed, notes, max_context_len, time_window, overflow, delta on every n-th call, group chords)
    "crop": (41, 140, 48, 0.5, 0.1, 3, True),          # context cropped at bar boundaries several times, delta embeddings
    "long_context": (42, 60, 512, 0.8, 0.05, 0, True),  # never cropped: caches reused and cut across calls
    "single_notes": (43, 40, 32, 0.3, 0.1, 2, False),   # group_chord_notes=False
}


def main():
    cfg, model, kw = build(dict(preset="tiny"), seed=3)
    model.eval()
    d_ctx, d_style = model.perf_decoder.model.context_emb_dim, model.perf_decoder.model.style_emb_dim
    out = {"vocab": np.array(repr(SMALL_VOCAB)), "weights_seed": np.array(3)}
    for name, (seed, n, ctx_len, window, overflow, delta_every, group) in SCENARIOS.items():
        piece = make_piece(seed, n, SMALL_VOCAB)
        dataset = make_dataset(SMALL_VOCAB, [piece])
        g = torch.Generator().manual_seed(seed)
        score_emb = torch.randn(n + 2, d_ctx, generator=g)               # stand-ins for the encoder outputs (inputs of the loop)
        perf_emb = torch.randn(n + 2, d_style, generator=g)
        delta = torch.randn(d_style, generator=g) * 0.3
        gen = ScorePerformerGenerator(model, dataset, MixedLMScorePerformanceCollator(**COLLATOR), FakeMessenger(SMALL_VOCAB), device="cpu")
        gen.prepare_performance_notes(0, score_embeddings=score_emb.clone(), perf_embeddings=perf_emb.clone())
        out[f"{name}/cfg"] = np.array(repr(dict(max_context_len=ctx_len, time_window=window, time_window_overflow=overflow,
                                                delta_every=delta_every, group_chord_notes=group)))
        out[f"{name}/piece"], out[f"{name}/score_emb"], out[f"{name}/perf_emb"], out[f"{name}/delta"] = piece, score_emb.numpy(), perf_emb.numpy(), delta.numpy()
        out[f"{name}/notes"] = gen.perf_data.notes.numpy()
        t, call = 0.0, 0
        while not gen.perf_data.reached_eos and call < 400:
            use_delta = delta_every and call % delta_every == 0
            seq, messages = gen.generate_performance_notes(
                start_time=t, time_window=window, time_window_overflow=overflow, delta_embedding=delta.clone() if use_delta else None,
                max_context_len=ctx_len, group_chord_notes=group, filter_logits_fn=top_k, filter_kwargs={"k": 1})
            out[f"{name}/call{call}/tokens"] = seq.numpy() if seq is not None else np.zeros((0, 12), np.int64)
            out[f"{name}/call{call}/messages"] = np.array(messages, np.float64).reshape(-1, 4)
            out[f"{name}/call{call}/cache_len"] = np.array(-1 if gen.perf_data.caches is None else gen.perf_data.caches.token_emb.shape[1])
            out[f"{name}/call{call}/predicted_notes"] = np.array(int(gen.predict_number_of_notes(start_time=t + window, time_window=window)))
            t += window
            call += 1
        out[f"{name}/calls"] = np.array(call)
        out[f"{name}/gen_seq"] = gen.perf_data.gen_seq.numpy()
        out[f"{name}/final_embeddings"] = gen.perf_data.embeddings.numpy()
        print(name, "calls", call, "generated", gen.perf_data.gen_seq.shape[0] - 1, "of", n, "reached_eos", gen.perf_data.reached_eos)
    path = os.path.join(OUT, "render_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
