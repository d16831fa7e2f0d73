"""GPU: `Attention` with the reference class's options beyond the shipped recipes through the HIP kernels -- learned memory keys /
values (`num_mem_kv`) and head widths below 64 (zero-padded on the weights, the kernels stay 64 wide) -- against the REFERENCE module's
own outputs and gradients (tests/golden/memkv.npz, units.npz; generators under oracle/refimport/).  bf16 GEMM operands: errors are
judged against each tensor's scale, as in tests/test_units_gpu.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
M = np.load(os.path.join(GOLD, "memkv.npz"))
U8 = np.load(os.path.join(GOLD, "units.npz"))
DIM, HEADS, MEM = 128, 2, 4


def rel(got, want):
    want = torch.from_numpy(np.asarray(want))
    return float((got.detach().float().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def _run(att, dev):
    x = torch.from_numpy(M["x"]).to(dev).requires_grad_(True)
    out = att(x, mask=torch.from_numpy(M["mask"]).to(dev))[0]
    (out.float() * torch.from_numpy(M["w"]).to(dev)).sum().backward()
    return x, out


@pytest.mark.parametrize("causal", [0, 1])
@pytest.mark.parametrize("dh", [64, 32])
def test_attention_with_memory_key_values_matches_the_reference(dev, causal, dh):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=DIM, dim_head=dh, heads=HEADS, causal=bool(causal), num_mem_kv=MEM, alibi_pos_bias=True, alibi_learned=True).eval()
    att.load_state_dict(filled_state_dict(att, seed=11))
    att.to(dev)
    x, out = _run(att, dev)
    tag = f"mem/c{causal}_d{dh}/"
    assert rel(out, M[tag + "out"]) <= 0.02
    assert rel(x.grad, M[tag + "dx"]) <= 0.03
    named = dict(att.named_parameters())
    for name, tol in (("mem_k", 0.04), ("mem_v", 0.03), ("to_k.weight", 0.04), ("rel_pos.learned_logslopes", 0.08)):
        assert named[name].grad is not None, name
        assert rel(named[name].grad, M[tag + "d_" + name]) <= tol, (name, rel(named[name].grad, M[tag + "d_" + name]))
    # the intermediates carry the memories in front of the sequence's keys, in the reference's [b, h, j, d] layout
    with torch.no_grad():
        inter = att(x.detach(), mask=torch.from_numpy(M["mask"]).to(dev))[1]
    assert tuple(inter.keys.shape) == (2, HEADS, MEM + 40, dh)


@pytest.mark.parametrize("mqa", [0, 1])
def test_attention_with_32_wide_heads_matches_the_reference(dev, mqa):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=DIM, dim_head=32, heads=HEADS, causal=True, one_kv_head=bool(mqa), alibi_pos_bias=True, alibi_learned=True).eval()
    att.load_state_dict(filled_state_dict(att, seed=12))
    assert tuple(att.to_q.weight.shape) == (HEADS * 32, DIM)             # parameters keep the reference's shapes
    att.to(dev)
    x, out = _run(att, dev)
    tag = f"narrow/m{mqa}/"
    assert rel(out, M[tag + "out"]) <= 0.02
    assert rel(x.grad, M[tag + "dx"]) <= 0.03
    named = dict(att.named_parameters())
    for name in ("to_q.weight", "to_out.weight"):
        assert rel(named[name].grad, M[tag + "d_" + name]) <= 0.04, name


@pytest.mark.parametrize("c", [0, 1])
@pytest.mark.parametrize("m", [0, 1])
@pytest.mark.parametrize("l", [0, 1])
def test_attention_with_8_wide_heads_matches_the_reference(dev, c, m, l):
    """The reference module's own fixture at dim = 32, four heads of width 8, nine positions (tests/golden/units.npz)."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=32, dim_head=8, heads=4, causal=bool(c), one_kv_head=bool(m), alibi_pos_bias=True, alibi_learned=bool(l)).eval()
    att.load_state_dict(filled_state_dict(att, seed=7))
    att.to(dev)
    with torch.no_grad():
        out = att(torch.from_numpy(U8["attn/x"]).to(dev), mask=torch.from_numpy(U8["attn/mask"]).to(dev))[0]
    assert rel(out, U8[f"attn/c{c}_m{m}_l{l}"]) <= 0.02
