"""Module-level vectors for the options of the reference `Attention` beyond the shipped recipes, from the REFERENCE class itself
(authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_memkv

`num_mem_kv` = 4 learned memory keys / values per head (modules/transformer/attention.py:98-101,146-153) x causal x head width
(64 and 32), multi-head K/V, learned ALiBi slopes, a padded sequence: the output and, for loss = sum(out * w), the gradients of the
input, of both memories and of the key projection.  Plus two plain narrow-head cases (dim_head = 32, no memories; multi-query and
multi-head) with their input gradients.  Weights from oracle.weights.  tests/golden/memkv.npz; data only.
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
from scoreperformer.modules.transformer import Attention  # noqa: E402  (the reference)

from oracle.weights import filled_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
DIM, HEADS, N, MEM = 128, 2, 40, 4


def run(att, x, mask, w, names):
    xr = x.clone().requires_grad_(True)
    out = att(xr, mask=mask)[0]
    (out * w).sum().backward()
    res = {"out": out.detach().numpy(), "dx": xr.grad.numpy()}
    named = dict(att.named_parameters())
    for n in names:
        res["d_" + n] = named[n].grad.numpy()
    return res


def main():
    fix = {}
    g = torch.Generator().manual_seed(23)
    x = torch.randn(2, N, DIM, generator=g)
    w = torch.randn(2, N, DIM, generator=g)
    mask = torch.ones(2, N, dtype=torch.bool); mask[1, 31:] = False
    fix.update({"x": x.numpy(), "w": w.numpy(), "mask": mask.numpy()})
    for causal in (False, True):
        for dh in (64, 32):
            att = Attention(dim=DIM, dim_head=dh, heads=HEADS, causal=causal, one_kv_head=False, num_mem_kv=MEM, alibi_pos_bias=True,
                            alibi_learned=True).eval()
            att.load_state_dict(filled_state_dict(att, seed=11))
            for k, v in run(att, x, mask, w, ["mem_k", "mem_v", "to_k.weight", "rel_pos.learned_logslopes"]).items():
                fix[f"mem/c{int(causal)}_d{dh}/{k}"] = v
    for mqa in (False, True):
        att = Attention(dim=DIM, dim_head=32, heads=HEADS, causal=True, one_kv_head=mqa, alibi_pos_bias=True, alibi_learned=True).eval()
        att.load_state_dict(filled_state_dict(att, seed=12))
        for k, v in run(att, x, mask, w, ["to_q.weight", "to_out.weight"]).items():
            fix[f"narrow/m{int(mqa)}/{k}"] = v
    path = os.path.join(OUT, "memkv.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, len(fix), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
