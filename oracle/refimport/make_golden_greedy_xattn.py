"""Golden greedy render of the REFERENCE with a CROSS-ATTENDING decoder (context_emb_mode='attention': decoder layer blocks
('a','c','f'), modules/transformer/transformer.py:92-93,201), authoring container only:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_greedy_xattn

Cached `unmask_tokens` (wrappers.py:325-407) on a 40-note window whose score has 6 padded positions (context_mask), eval mode,
top-k k=1.  tests/golden/tiny_greedy_xattn.npz: inputs, the context mask, the reference's tokens and encoder outputs.  Data only.
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
from oracle.refimport import stubs  # noqa: E402

stubs.install()
warnings.filterwarnings("ignore")
from scoreperformer.models import ScorePerformer  # noqa: E402  (the reference)
from scoreperformer.modules.sampling import top_k  # noqa: E402

from oracle.refimport.make_golden import SMALL_VOCAB  # noqa: E402
from oracle.weights import filled_state_dict  # noqa: E402
from scoreperformer_amd.synthetic import model_config, synthetic_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
CFG = dict(preset="tiny", context_emb_mode="attention", num_tokens=SMALL_VOCAB)


def main():
    model = ScorePerformer.init(copy.deepcopy(model_config(**CFG)))
    model.load_state_dict(filled_state_dict(model, seed=4), strict=True)
    model.eval()
    batch = synthetic_batch(1, 40, num_tokens=SMALL_VOCAB, seed=23)
    score_mask = batch["score_mask"].clone()
    score_mask[:, 34:] = False                      # the score is shorter than the window: 6 padded context positions
    score = batch["score"] * score_mask[..., None]
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=score, score_mask=score_mask,
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
        tokens = batch["masked_perf"].clone()
        tokens[:, 0] = batch["perf"][:, 0]
        out = model.perf_decoder.unmask_tokens(tokens, batch["masked_perf"], context=enc.score_embeddings, context_mask=score_mask,
                                               style_embeddings=enc.perf_embeddings, filter_logits_fn=top_k, filter_kwargs={"k": 1},
                                               disable_tqdm=True)
    fix = {f"in/{k}": v.numpy() for k, v in batch.items()}
    fix["in/score"], fix["in/score_mask"] = score.numpy(), score_mask.numpy()
    fix["in/tokens"], fix["out/tokens"] = tokens.numpy(), out.numpy()
    fix["out/score_embeddings"], fix["out/perf_embeddings"] = enc.score_embeddings.numpy(), enc.perf_embeddings.numpy()
    path = os.path.join(OUT, "tiny_greedy_xattn.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, "tokens", out.shape, "changed", int((out != tokens).sum()), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
