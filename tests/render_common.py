"""Shared by the CPU (oracle) and GPU (product) render-loop tests: the golden scenarios of tests/golden/render_loop.npz."""
import ast
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "render_loop.npz")
COLLATOR = dict(pad_token_id=0, pad_to_multiple_of=1, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
                mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9])
MASK_DIMS = [3, 5, 10, 11]


def load():
    z = np.load(GOLD, allow_pickle=False)
    vocab = ast.literal_eval(str(z["vocab"]))
    names = sorted({k.split("/")[0] for k in z.files if "/" in k})
    scen = {}
    for name in names:
        cfg = ast.literal_eval(str(z[f"{name}/cfg"]))
        calls = [dict(tokens=z[f"{name}/call{i}/tokens"], messages=z[f"{name}/call{i}/messages"], cache_len=int(z[f"{name}/call{i}/cache_len"]),
                      predicted_notes=int(z[f"{name}/call{i}/predicted_notes"])) for i in range(int(z[f"{name}/calls"]))]
        scen[name] = dict(cfg=cfg, piece=z[f"{name}/piece"], score_emb=z[f"{name}/score_emb"], perf_emb=z[f"{name}/perf_emb"],
                          delta=z[f"{name}/delta"], notes=z[f"{name}/notes"], calls=calls, gen_seq=z[f"{name}/gen_seq"],
                          final_embeddings=z[f"{name}/final_embeddings"])
    return vocab, int(z["weights_seed"]), scen
