"""Module-level vectors from the REFERENCE classes at head dim 64 (the width the HIP attention kernels are built for), authoring
container only:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_units64

`Attention` (causal x multi-query x learned slopes, plus two cross-attention cases), `FeedForward` (GLU x SiLU/GELU) and
`AdaptiveLayerNorm`, eval mode, fixed inputs with a padded sequence; weights from oracle.weights.  tests/golden/units64.npz; data only.
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
from scoreperformer.modules.layers import AdaptiveLayerNorm  # noqa: E402  (the reference)
from scoreperformer.modules.transformer import Attention, FeedForward  # noqa: E402

from oracle.weights import filled_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
DIM, HEADS, N, NC = 128, 2, 40, 56


def main():
    fix = {}
    g = torch.Generator().manual_seed(15)
    x = torch.randn(2, N, DIM, generator=g)
    ctx = torch.randn(2, NC, DIM, generator=g)
    mask = torch.ones(2, N, dtype=torch.bool); mask[1, 29:] = False
    cmask = torch.ones(2, NC, dtype=torch.bool); cmask[0, 50:] = False
    fix.update({"x": x.numpy(), "ctx": ctx.numpy(), "mask": mask.numpy(), "cmask": cmask.numpy()})
    for causal in (False, True):
        for mqa in (False, True):
            for learned in (False, True):
                att = Attention(dim=DIM, dim_head=64, heads=HEADS, causal=causal, one_kv_head=mqa, alibi_pos_bias=True,
                                alibi_learned=learned).eval()
                att.load_state_dict(filled_state_dict(att, seed=7))
                with torch.no_grad():
                    fix[f"attn/c{int(causal)}_m{int(mqa)}_l{int(learned)}"] = att(x, mask=mask)[0].numpy()
    for mqa in (False, True):   # cross-attention over a longer, padded context (decoder layer type 'c')
        att = Attention(dim=DIM, dim_head=64, heads=HEADS, causal=False, one_kv_head=mqa, alibi_pos_bias=True, alibi_learned=True).eval()
        att.load_state_dict(filled_state_dict(att, seed=7))
        with torch.no_grad():
            fix[f"xattn/m{int(mqa)}"] = att(x, context=ctx, mask=mask, context_mask=cmask)[0].numpy()
    for glu in (False, True):
        for swish in (False, True):
            ff = FeedForward(dim=DIM, mult=2, glu=glu, swish=swish).eval()
            ff.load_state_dict(filled_state_dict(ff, seed=8))
            with torch.no_grad():
                fix[f"ff/g{int(glu)}_s{int(swish)}"] = ff(x).numpy()
    ada = AdaptiveLayerNorm(DIM, 16)
    ada.load_state_dict(filled_state_dict(ada, seed=9))
    cond = torch.randn(2, N, 16, generator=g)
    fix["ada/cond"] = cond.numpy()
    with torch.no_grad():
        fix["ada/out"] = ada(x, condition=cond).numpy()
    path = os.path.join(OUT, "units64.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, len(fix), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
