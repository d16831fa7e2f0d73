"""GPU: the mirrored modules themselves -- `Attention` (self, causal, multi-query, learned slopes, cross), `FeedForward` (GLU x
SiLU / GELU) and `AdaptiveLayerNorm` -- through the HIP kernels against the REFERENCE modules' outputs on the same weights and inputs
(tests/golden/units64.npz, oracle/refimport/make_golden_units64.py).  bf16 GEMM operands: errors are judged against the output scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
U = np.load(os.path.join(os.path.dirname(__file__), "golden", "units64.npz"))


def close(got, want, tol=0.02):
    want = torch.from_numpy(want)
    err = (got.detach().float().cpu() - want).abs().max() / want.abs().max()
    assert float(err) <= tol, (float(err), tol)


def on(dev, name):
    return torch.from_numpy(U[name]).to(dev)


@pytest.mark.parametrize("c", [0, 1])
@pytest.mark.parametrize("m", [0, 1])
@pytest.mark.parametrize("l", [0, 1])
def test_attention_module_matches_the_reference(dev, c, m, l):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=128, dim_head=64, heads=2, causal=bool(c), one_kv_head=bool(m), alibi_pos_bias=True, alibi_learned=bool(l)).eval()
    att.load_state_dict(filled_state_dict(att, seed=7))
    att.to(dev)
    with torch.no_grad():
        out = att(on(dev, "x"), mask=on(dev, "mask"))[0]
    close(out, U[f"attn/c{c}_m{m}_l{l}"])


@pytest.mark.parametrize("m", [0, 1])
def test_cross_attention_module_matches_the_reference(dev, m):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=128, dim_head=64, heads=2, causal=False, one_kv_head=bool(m), alibi_pos_bias=True, alibi_learned=True).eval()
    att.load_state_dict(filled_state_dict(att, seed=7))
    att.to(dev)
    with torch.no_grad():
        out = att(on(dev, "x"), context=on(dev, "ctx"), mask=on(dev, "mask"), context_mask=on(dev, "cmask"))[0]
    close(out, U[f"xattn/m{m}"])


@pytest.mark.parametrize("glu", [0, 1])
@pytest.mark.parametrize("swish", [0, 1])
def test_feed_forward_module_matches_the_reference(dev, glu, swish):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import FeedForward
    ff = FeedForward(dim=128, mult=2, glu=bool(glu), swish=bool(swish)).eval()
    ff.load_state_dict(filled_state_dict(ff, seed=8))
    ff.to(dev)
    with torch.no_grad():
        out = ff(on(dev, "x"))
    close(out, U[f"ff/g{glu}_s{swish}"])


def test_adaptive_layer_norm_module_matches_the_reference(dev):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.layers import AdaptiveLayerNorm
    ada = AdaptiveLayerNorm(128, 16)
    ada.load_state_dict(filled_state_dict(ada, seed=9))
    ada.to(dev)
    with torch.no_grad():
        out = ada(on(dev, "x"), condition=on(dev, "ada/cond"), out_fp32=True)
    close(out, U["ada/out"], tol=1e-2)   # (gamma | beta) = Linear(condition) runs with bf16 operands
