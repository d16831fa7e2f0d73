"""CPU: the oracle's attention with the reference class's options beyond the shipped recipes -- learned memory keys / values
(`num_mem_kv`, modules/transformer/attention.py:98-101,146-153) and head widths below 64 -- against vectors written by the REFERENCE
module itself (tests/golden/memkv.npz, oracle/refimport/make_golden_memkv.py): output and gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.weights import filled_state_dict

M = np.load(os.path.join(os.path.dirname(__file__), "golden", "memkv.npz"))
DIM, HEADS, MEM = 128, 2, 4


def _leaves(att, seed):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in filled_state_dict(att, seed=seed).items()}


def _run(sd, causal, heads=HEADS):
    x = torch.from_numpy(M["x"]).clone().requires_grad_(True)
    out = ref_cpu.attention(sd, "", x, heads=heads, causal=causal, alibi=True, mask=torch.from_numpy(M["mask"]))
    (out * torch.from_numpy(M["w"])).sum().backward()
    return x, out


@pytest.mark.parametrize("causal", [0, 1])
@pytest.mark.parametrize("dh", [64, 32])
def test_oracle_attention_with_memory_key_values_matches_the_reference_module(causal, dh):
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=DIM, dim_head=dh, heads=HEADS, causal=bool(causal), num_mem_kv=MEM, alibi_pos_bias=True, alibi_learned=True)
    assert tuple(att.mem_k.shape) == (HEADS, MEM, dh) and "mem_k" in att.state_dict()      # the reference's parameter layout
    sd = _leaves(att, 11)
    x, out = _run(sd, bool(causal))
    tag = f"mem/c{causal}_d{dh}/"
    np.testing.assert_allclose(out.detach().numpy(), M[tag + "out"], atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), M[tag + "dx"], atol=5e-6, rtol=1e-4)
    for name in ("mem_k", "mem_v", "to_k.weight", "rel_pos.learned_logslopes"):
        np.testing.assert_allclose(sd[name].grad.numpy(), M[tag + "d_" + name], atol=2e-5, rtol=1e-4, err_msg=name)


@pytest.mark.parametrize("mqa", [0, 1])
def test_oracle_attention_with_narrow_heads_matches_the_reference_module(mqa):
    from scoreperformer_amd.modules.transformer import Attention
    att = Attention(dim=DIM, dim_head=32, heads=HEADS, causal=True, one_kv_head=bool(mqa), alibi_pos_bias=True, alibi_learned=True)
    sd = _leaves(att, 12)
    x, out = _run(sd, True)
    tag = f"narrow/m{mqa}/"
    np.testing.assert_allclose(out.detach().numpy(), M[tag + "out"], atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), M[tag + "dx"], atol=5e-6, rtol=1e-4)
    for name in ("to_q.weight", "to_out.weight"):
        np.testing.assert_allclose(sd[name].grad.numpy(), M[tag + "d_" + name], atol=2e-5, rtol=1e-4, err_msg=name)


def test_undefined_option_combinations_fail_loudly():
    from scoreperformer_amd.modules.transformer import Attention
    with pytest.raises(NotImplementedError):
        Attention(dim=DIM, dim_head=128, heads=2)                         # wider than the kernels' head
    with pytest.raises(NotImplementedError):
        Attention(dim=DIM, heads=2, max_attend=16)                        # the reference's window mask hides everything but the far future
    with pytest.raises(NotImplementedError):
        Attention(dim=DIM, heads=2, num_mem_kv=4, one_kv_head=True)       # the reference fails in torch.cat here
