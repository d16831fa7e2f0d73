"""GPU check of a VARIANT build (not part of tests/: the kernels are not in libspn.so since round 6).

    python tools/build_variant.py gemm.hip ow2_spn.so -DSPN_GEMM_OW_VARIANT=2      (LDS-DMA operands;  =4: register-staged operands)
    SPN_LIB=tools/_bin/ow2_spn.so python -m pytest tools/variants/check_gemm_ow.py -q -p no:cacheprovider --rootdir tests -c /dev/null

The one-wave-per-SIMD GEMM kernels (tools/variants/gemm_ow_kernels.inc: four waves per 256x256 tile, 128x128 wave tiles, K tile 32)
against fp32 torch products of the same bf16 operands AND bit for bit against the ping-pong kernel (same 16-wide k steps in the same
order, same split-K plan, same epilogue: every output element sees the same sequence of roundings).  In a variant build the dispatch
takes the one-wave kernel by default (`gemm_variant` 0) and the ping-pong kernel with `gemm_variant` 9."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


@pytest.fixture()
def knob():
    from scoreperformer_amd import lib
    if "SPN_LIB" not in os.environ:
        pytest.skip("needs a variant build: SPN_LIB=tools/_bin/<variant>.so")
    before = lib.get_tuning("gemm_variant")
    yield lambda v: lib.set_tuning("gemm_variant", 9.0 if v == 0 else 0.0)      # 0: the ping-pong kernel; else: the variant's kernel
    lib.set_tuning("gemm_variant", before)


@pytest.mark.parametrize("kv", [1])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 1088), (1024, 512, 64 * 37), (4088, 640, 512), (392, 648, 320),
                                   (136, 1160, 256), (2048, 512, 4096)])
def test_ow_kernel_layouts_edges_and_bit_identity(dev, knob, kv, ta, tb, M, N, K):
    from scoreperformer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = torch.randn(K, N, generator=g).to(dev).bfloat16()
    ref = a.float() @ b.float()
    a_store = a.t().contiguous() if ta else a
    b_store = b if tb else b.t().contiguous()
    outs = {}
    for v in (0, kv):
        knob(v)
        outs[v] = (ops.gemm(a_store, b_store, ta=ta, tb=tb, out_dtype=torch.float32), ops.gemm(a_store, b_store, ta=ta, tb=tb, out_dtype=torch.bfloat16))
    assert rel_err(outs[kv][0], ref) < 2e-3 and rel_err(outs[kv][1], ref) < 1e-2
    assert torch.equal(outs[0][0], outs[kv][0]) and torch.equal(outs[0][1], outs[kv][1])


@pytest.mark.parametrize("kv", [1])
def test_ow_kernel_epilogues_and_split_k(dev, knob, kv):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 768, 512, 320
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = torch.randn(N, K, generator=g).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.3).to(dev)
    ref = res + mask[:, None] * (0.5 * (a.float() @ w.float().t()) + bias)
    T = 32768
    dy = torch.randn(T, 512, generator=g).to(dev).bfloat16()
    x = torch.randn(T, 256, generator=g).to(dev).bfloat16()
    got = {}
    for v in (0, kv):
        knob(v)
        out = ops.gemm(a, w, out_dtype=torch.float32, bias=bias, residual=res, rowmask=mask, alpha=0.5)
        out16 = ops.gemm(a, w, out_dtype=torch.bfloat16, bias=bias)
        dw = ops.gemm(dy, x, ta=True, tb=True, out_dtype=torch.float32)           # split-K through the workspace
        acc = torch.ones(512, 256, device=dev)
        ops.gemm(dy, x, ta=True, tb=True, out=acc, accumulate=True)
        got[v] = (out, out16, dw, acc)
    out, out16, dw, acc = got[kv]
    assert rel_err(out, ref) < 2e-3
    assert rel_err(out16, a.float() @ w.float().t() + bias) < 1e-2
    assert rel_err(dw, dy.float().t() @ x.float()) < 2e-3
    assert rel_err(acc, 1 + dy.float().t() @ x.float()) < 2e-3
    for p, q in zip(got[0], got[kv]):
        assert torch.equal(p, q)


@pytest.mark.parametrize("kv", [1])
def test_ow_kernel_random_shape_screen(knob, kv):
    """tools/stress_gemm.py (random shapes, every layout, full-tensor comparison) with the kernel forced wherever it is eligible."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stress_gemm", os.path.join(root, "tools", "stress_gemm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    knob(kv)
    assert mod.run(60, 11, verbose=False) == 0
