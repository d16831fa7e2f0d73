"""GPU parity tests of the individual HIP kernels (through the C-ABI) against fp32 PyTorch math on the same inputs.

Tolerances: operands are bf16, accumulation fp32.  A bf16 result is compared at rtol 2^-7 on the tensor scale.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1000, 264, 520), (264, 1000, 2048),
                                   (512, 256, 256), (256, 768, 1088), (1024, 512, 64 * 37),   # whole 256x256x64 tiles
                                   (4088, 640, 512), (392, 648, 320), (136, 1160, 256)])      # ping-pong kernel with M / N edge tiles
def test_gemm_layouts(dev, ta, tb, M, N, K):
    from scoreperformer_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = torch.randn(K, N, generator=g).to(dev).bfloat16()  # asymmetric by construction
    ref = a.float() @ b.float()
    a_store = a.t().contiguous() if ta else a
    b_store = b if tb else b.t().contiguous()
    out = ops.gemm(a_store, b_store, ta=ta, tb=tb, out_dtype=torch.float32)
    assert rel_err(out, ref) < 2e-3
    out16 = ops.gemm(a_store, b_store, ta=ta, tb=tb, out_dtype=torch.bfloat16)
    assert rel_err(out16, ref) < 1e-2


def test_gemm_epilogue(dev):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 192, 136
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = torch.randn(N, K, generator=g).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.3).to(dev)
    ref = res + mask[:, None] * (0.5 * (a.float() @ w.float().t()) + bias)
    out = ops.gemm(a, w, out_dtype=torch.float32, bias=bias, residual=res, rowmask=mask, alpha=0.5)
    assert rel_err(out, ref) < 2e-3
    acc = torch.ones(M, N, device=dev)
    ops.gemm(a, w, out=acc, accumulate=True)
    assert rel_err(acc, 1 + a.float() @ w.float().t()) < 2e-3
    # strided views (slices of wider buffers)
    wide = torch.randn(M, 3 * K, generator=g).to(dev).bfloat16()
    outw = torch.zeros(M, 2 * N, device=dev, dtype=torch.bfloat16)
    ops.gemm(wide[:, K:2 * K], w, out=outw[:, N:])
    assert rel_err(outw[:, N:], wide[:, K:2 * K].float() @ w.float().t()) < 1e-2
    assert outw[:, :N].abs().max().item() == 0


def test_gemm_big_tile_epilogue_and_splitk(dev):
    """Shapes that take the 256x256 ping-pong kernel: fused epilogue, accumulate, and the split-K weight-gradient form."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 768, 512, 320
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = torch.randn(N, K, generator=g).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.3).to(dev)
    ref = res + mask[:, None] * (0.5 * (a.float() @ w.float().t()) + bias)
    out = ops.gemm(a, w, out_dtype=torch.float32, bias=bias, residual=res, rowmask=mask, alpha=0.5)
    assert rel_err(out, ref) < 2e-3
    out16 = ops.gemm(a, w, out_dtype=torch.bfloat16, bias=bias)
    assert rel_err(out16, a.float() @ w.float().t() + bias) < 1e-2
    # dW = dY^T X over many tokens (split-K with atomics), accumulated onto an existing gradient
    T = 16384
    dy = torch.randn(T, 512, generator=g).to(dev).bfloat16()
    x = torch.randn(T, 256, generator=g).to(dev).bfloat16()
    ref = dy.float().t() @ x.float()
    dw = ops.gemm(dy, x, ta=True, tb=True, out_dtype=torch.float32)
    assert rel_err(dw, ref) < 2e-3
    acc = torch.ones(512, 256, device=dev)
    ops.gemm(dy, x, ta=True, tb=True, out=acc, accumulate=True)
    assert rel_err(acc, 1 + ref) < 2e-3


@pytest.mark.parametrize("D", [128, 320, 512, 1280, 1536])
@pytest.mark.parametrize("mode", ["affine", "ada", "plain"])
def test_layernorm(dev, D, mode):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(D)
    T = 77
    x = (torch.randn(T, D, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev) if mode == "affine" else None
    beta = (0.1 * torch.randn(D, generator=g)).to(dev) if mode == "affine" else None
    gb = torch.cat([1 + 0.1 * torch.randn(T, D, generator=g), 0.1 * torch.randn(T, D, generator=g)], -1).to(dev) \
        if mode == "ada" else None
    dy = torch.randn(T, D, generator=g).to(dev).bfloat16()

    xr = x.clone().requires_grad_(True)
    leaves = [xr]
    if mode == "affine":
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(xr, (D,), gr, br)
        leaves += [gr, br]
    elif mode == "ada":
        gbr = gb.clone().requires_grad_(True)
        ref = gbr[:, :D] * torch.nn.functional.layer_norm(xr, (D,)) + gbr[:, D:]
        leaves += [gbr]
    else:
        ref = torch.nn.functional.layer_norm(xr, (D,))
    ref.backward(dy.float())

    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, gb, out_dtype=torch.float32)
    assert rel_err(y, ref) < 1e-5
    y16, _, _ = ops.layernorm_fwd(x, gamma, beta, gb, out_dtype=torch.bfloat16)
    assert rel_err(y16, ref) < 1e-2
    dgamma = torch.zeros(D, device=dev) if mode == "affine" else None
    dbeta = torch.zeros(D, device=dev) if mode == "affine" else None
    dres = torch.randn(T, D, generator=g).to(dev)
    dx, dgb = ops.layernorm_bwd(x, dy, gamma, gb, mean, rstd, dres=dres, dgamma=dgamma, dbeta=dbeta, want_dgb=mode == "ada")
    assert rel_err(dx - dres, xr.grad) < 1e-4
    if mode == "affine":
        assert rel_err(dgamma, leaves[1].grad) < 1e-4
        assert rel_err(dbeta, leaves[2].grad) < 1e-4
    if mode == "ada":
        assert rel_err(dgb, leaves[1].grad) < 1e-2


def attn_reference(q, k, v, kmask, slopes, causal, scale):
    """fp32 restatement of attend.py:58-126 + attention.py:162-197 (same as oracle/ref_cpu.attention core)."""
    b, nq, h, dh = q.shape
    nk, kvh = k.shape[1], k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).expand(b, h, nk, dh)
    vf = v.float().permute(0, 2, 1, 3).expand(b, h, nk, dh)
    dots = qf @ kf.transpose(-1, -2) * scale
    ii = torch.arange(nk - nq, nk, device=q.device)
    jj = torch.arange(nk, device=q.device)
    dist = (jj[None, :] - ii[:, None])
    if slopes is not None:
        dots = dots - slopes.view(1, h, 1, 1) * dist.abs().float()
    allowed = torch.ones(b, 1, nq, nk, dtype=torch.bool, device=q.device)
    if kmask is not None:
        allowed = allowed & kmask[:, None, None, :]
    if causal:
        allowed = allowed & (dist <= 0)
    dots = torch.where(allowed, dots, torch.full_like(dots, -1.7014118e38))
    p = dots.softmax(-1)
    return (p @ vf).permute(0, 2, 1, 3), dots.logsumexp(-1)


@pytest.mark.parametrize("mqa", [True, False])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("b,h,nq,nk", [(2, 4, 128, 128), (3, 8, 200, 200), (2, 2, 47, 48), (1, 8, 300, 333),
                                        # block-order remaps (attention_common.h): batch a multiple of 8 (XCD-per-batch order, causal
                                        # longest-first), 16 query tiles with batch 1 (mirror-pair causal order), nq != nk with batch 8
                                        (8, 2, 256, 256), (16, 2, 384, 384), (1, 2, 2048, 2048), (8, 4, 300, 333)])
def test_attention_fwd_bwd(dev, mqa, causal, b, h, nq, nk):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(b * 1000 + h * 100 + nq + nk + int(mqa) * 7 + int(causal))
    kvh = 1 if mqa else h
    width = h * 64 + 2 * kvh * 64
    qkv = torch.randn(b, nk, width, generator=g).to(dev).bfloat16()  # fused projection buffer
    if nq != nk:
        q = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()
    else:
        q = qkv[..., :h * 64].unflatten(-1, (h, 64))
    k = qkv[..., h * 64:h * 64 + kvh * 64].unflatten(-1, (kvh, 64))
    v = qkv[..., h * 64 + kvh * 64:].unflatten(-1, (kvh, 64))
    lens = torch.randint(nk // 2, nk + 1, (b,), generator=g)
    kmask = (torch.arange(nk)[None, :] < lens[:, None]).to(dev)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / h) for i in range(h)], device=dev) * 1.3
    scale = 64 ** -0.5
    d_o = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()

    qr, kr, vr, sr = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True), \
        slopes.clone().requires_grad_(True)
    ref, ref_lse = attn_reference(qr, kr, vr, kmask, sr, causal, scale)
    ref.backward(d_o.float())

    o, lse = ops.attn_fwd(q, k, v, kmask=kmask, slopes=slopes, causal=causal, scale=scale)
    assert rel_err(o, ref) < 2e-2
    assert (lse - ref_lse).abs().max().item() < 2e-2

    dqkv = torch.zeros(b, nk, width, device=dev, dtype=torch.bfloat16)
    dq = torch.zeros_like(q) if nq != nk else dqkv[..., :h * 64].unflatten(-1, (h, 64))
    dk = dqkv[..., h * 64:h * 64 + kvh * 64].unflatten(-1, (kvh, 64))
    dv = dqkv[..., h * 64 + kvh * 64:].unflatten(-1, (kvh, 64))
    dslope = ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, slopes=slopes, causal=causal,
                          scale=scale, want_dslope=True)
    assert rel_err(dq, qr.grad) < 3e-2
    assert rel_err(dk, kr.grad) < 3e-2
    assert rel_err(dv, vr.grad) < 3e-2
    assert rel_err(dslope, sr.grad) < 3e-2


@pytest.mark.parametrize("band", [0.0, 30.0])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("odd", [False, True])
def test_attention_query_mask_drops_padding_rows_and_nothing_else(dev, odd, cross, causal, p_drop, band):
    """qmask (the module's output mask, attention.py:216-218): rows with 0 come back as zeros with a dead lse and carry no gradient in
    either direction, whatever dO holds there; every other row, and dK / dV / d slope, are those of the same call without qmask and
    with dO zeroed on the padding rows -- bit for bit when the ALiBi band is off (with it, the padding rows no longer widen the bound
    of the tile they share with live rows: differences below the band's own 2^-19)."""
    from scoreperformer_amd import ops
    b, h, nk = 5, 4, (333 if odd else 640)   # odd: mask rows off the 16-byte grid (no vector scans), ragged last tiles
    nq = (300 if odd else 384) if cross else nk
    g = torch.Generator().manual_seed(17 + int(cross) + 2 * int(causal))
    q = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()
    k = torch.randn(b, nk, 1, 64, generator=g).to(dev).bfloat16()
    v = torch.randn(b, nk, 1, 64, generator=g).to(dev).bfloat16()
    klens = torch.tensor([333, 200, 129, 64, 301]) if odd else torch.tensor([640, 400, 129, 64, 577])
    # whole blocks, whole tiles, waves and single rows of padding
    qlens = (torch.tensor([300, 200, 1, 128, 257]) if odd else torch.tensor([384, 200, 1, 128, 300])) if cross else klens
    kmask = (torch.arange(nk)[None, :] < klens[:, None]).to(dev)
    qmask = (torch.arange(nq)[None, :] < qlens[:, None]).to(dev)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / h) for i in range(h)], device=dev)
    d_o = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()
    d_o_zeroed = d_o * qmask[:, :, None, None]
    kw = dict(kmask=kmask, slopes=slopes, causal=causal, scale=0.125)

    def run(qm, dout):
        fw = ops.attn_fwd(q, k, v, qmask=qm, p_drop=p_drop, seed=5, **kw)
        o, lse = fw[0], fw[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dsl = ops.attn_bwd(q, k, v, o, dout, lse, dq=dq, dk=dk, dv=dv, qmask=qm, want_dslope=True, p_drop=p_drop,
                           dropbits=fw[2] if p_drop > 0 else None, **kw)
        return o, lse, dq, dk, dv, dsl

    ops.attn_set_band(band)
    try:
        o0, lse0, dq0, dk0, dv0, ds0 = run(None, d_o_zeroed)
        o1, lse1, dq1, dk1, dv1, ds1 = run(qmask, d_o)
    finally:
        ops.attn_set_band(30.0)
    dead = ~qmask
    assert o1[dead].abs().max().item() == 0.0 and dq1[dead].abs().max().item() == 0.0
    assert (lse1.transpose(1, 2)[dead] < -1e37).all()
    live = qmask
    if band == 0.0:
        assert torch.equal(o1[live], o0[live]) and torch.equal(lse1.transpose(1, 2)[live], lse0.transpose(1, 2)[live])
        assert torch.equal(dq1[live], dq0[live]) and torch.equal(dk1, dk0) and torch.equal(dv1, dv0)
    else:
        assert rel_err(o1[live], o0[live]) < 1e-3 and rel_err(dq1[live], dq0[live]) < 1e-3
        assert rel_err(dk1, dk0) < 1e-3 and rel_err(dv1, dv0) < 1e-3
    assert rel_err(ds1, ds0) < 1e-4   # float atomics over the blocks: order-dependent in the last bits


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("use_slopes,use_mask", [(True, False), (False, False), (True, True)])
def test_attention_fast_paths_and_rescale(dev, causal, use_slopes, use_mask):
    """Long rows exercise the mask-free LEFT/RIGHT tile classes; a spiked key forces the deferred-max rescale branch
    (running max jumps by far more than 2^8 in the middle of a row)."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(17 + int(causal) + 2 * int(use_slopes) + 4 * int(use_mask))
    b, h, n = 2, 4, 640
    qkv = torch.randn(b, n, (h + 2) * 64, generator=g)
    # spike: key 300 of batch 0 is strongly aligned with query 500 (and 100): raw score ~ 40 * 64 / 8
    qkv[0, 300, h * 64:(h + 1) * 64] = 5.0
    qkv[0, 500, :64] = 8.0
    qkv[0, 100, :64] = 8.0
    qkv = qkv.to(dev).bfloat16()
    q = qkv[..., :h * 64].unflatten(-1, (h, 64))
    k = qkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64))
    v = qkv[..., (h + 1) * 64:].unflatten(-1, (1, 64))
    kmask = None
    if use_mask:
        kmask = torch.ones(b, n, dtype=torch.bool)
        kmask[1, 450:] = False
        kmask[0, 130] = False
        kmask = kmask.to(dev)
    slopes = torch.tensor([0.5, 0.25, 0.05, 0.01], device=dev) if use_slopes else None
    scale = 64 ** -0.5
    d_o = torch.randn(b, n, h, 64, generator=g).to(dev).bfloat16()
    qr, kr, vr = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    sr = slopes.clone().requires_grad_(True) if use_slopes else None
    ref, ref_lse = attn_reference(qr, kr, vr, kmask, sr, causal, scale)
    ref.backward(d_o.float())
    o, lse = ops.attn_fwd(q, k, v, kmask=kmask, slopes=slopes, causal=causal, scale=scale)
    assert torch.isfinite(o.float()).all()
    assert rel_err(o, ref) < 2e-2
    assert (lse - ref_lse).abs().max().item() < 2e-2
    dqkv = torch.zeros_like(qkv)
    dq = dqkv[..., :h * 64].unflatten(-1, (h, 64))
    dk = dqkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64))
    dv = dqkv[..., (h + 1) * 64:].unflatten(-1, (1, 64))
    dslope = ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, slopes=slopes, causal=causal, scale=scale,
                          want_dslope=use_slopes)
    assert torch.isfinite(dqkv.float()).all()
    assert rel_err(dq, qr.grad) < 3e-2
    assert rel_err(dk, kr.grad) < 3e-2
    assert rel_err(dv, vr.grad) < 3e-2
    if use_slopes:
        assert rel_err(dslope, sr.grad) < 3e-2


@pytest.mark.parametrize("causal,masked", [(False, False), (True, False), (False, True)])
def test_attention_alibi_band_skipping_is_invisible(dev, causal, masked):
    """Tiles outside the ALiBi reach are skipped (forward, dQ, dK/dV): the result must equal the visit-everything run."""
    from scoreperformer_amd import ops
    b, n, h = 2, 1024, 8
    g = torch.Generator().manual_seed(17)
    qkv = (torch.randn(b, n, (h + 2) * 64, generator=g) * 1.5).to(dev).bfloat16()
    q, k, v = (qkv[..., :h * 64].unflatten(-1, (h, 64)), qkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)),
               qkv[..., (h + 1) * 64:].unflatten(-1, (1, 64)))
    slopes = torch.tensor([2.0 ** (-(i + 1)) for i in range(h)], device=dev)
    kmask = None
    if masked:
        kmask = torch.ones(b, n, dtype=torch.bool, device=dev)
        kmask[0, 700:] = False
        kmask[1, 333:] = False
    d_o = torch.randn(b, n, h, 64, generator=g).to(dev).bfloat16()
    res = []
    try:
        for thr in (0.0, 40.0, 30.0):   # off / below fp32 resolution / the shipped default
            ops.attn_set_band(thr)
            o, lse = ops.attn_fwd(q, k, v, kmask=kmask, slopes=slopes, causal=causal)
            dqkv = torch.zeros_like(qkv)
            dq, dk, dv = (dqkv[..., :h * 64].unflatten(-1, (h, 64)), dqkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)),
                          dqkv[..., (h + 1) * 64:].unflatten(-1, (1, 64)))
            dsl = ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, slopes=slopes, causal=causal, want_dslope=True)
            res.append((o.float(), lse, dqkv.float(), dsl))
    finally:
        ops.attn_set_band(30.0)
    (o0, l0, d0, s0), (o1, l1, d1, s1), (o2, l2, d2, s2) = res
    valid = slice(None) if kmask is None else kmask   # rows at padded positions are don't-care in the model, but checked too
    assert (o0 - o1).abs().max() <= 1e-6 * o0.abs().max()
    assert (l0 - l1).abs().max() <= 1e-5
    assert (d0 - d1).abs().max() <= 1e-6 * d0.abs().max()
    assert (s0 - s1).abs().max() <= 1e-5 * s0.abs().max()
    # the default threshold 2^-30: what is skipped sums to < 2^-19 of a row's normaliser (bf16 rounds the probabilities at 2^-9)
    assert (o0 - o2).abs().max() <= 1e-5 * o0.abs().max() and (l0 - l2).abs().max() <= 1e-4
    assert (d0 - d2).abs().max() <= 1e-5 * d0.abs().max() and (s0 - s2).abs().max() <= 1e-4 * s0.abs().max()


@pytest.mark.parametrize("causal", [False, True])
def test_attention_alibi_band_is_safe_on_trained_like_inputs(dev, causal):
    """The band is an approximation that ships switched on (threshold 2^-30, csrc/tuning.h): it must stay invisible on inputs unlike the
    random-normal ones above -- queries and keys that share a strong common direction (scores q.k * scale of 40 and more, the regime of a
    trained model's sharp heads), learned slopes from 0.5 down to 2^-10, one head with slope 0 and one with a NEGATIVE slope (both must
    disable skipping for that head), ragged key masks (rows whose own key is masked have no lower bound on their maximum and are never
    skipped).  Same bounds as test_attention_alibi_band_skipping_is_invisible."""
    from scoreperformer_amd import ops
    b, n, h = 2, 1024, 8
    g = torch.Generator().manual_seed(23)
    common = torch.randn(1, 1, 64, generator=g) * 2.0
    qh = torch.randn(b, n, h, 64, generator=g) * 2.5 + common[:, :, None, :]
    kh = torch.randn(b, n, 1, 64, generator=g) * 2.5 + common[:, :, None, :]
    # a few outlier keys / queries with 3x the norm, as attention sinks have
    kh[:, ::97] *= 3.0
    qh[:, ::131] *= 3.0
    vh = torch.randn(b, n, 1, 64, generator=g)
    qkv = torch.cat([qh.flatten(2), kh.flatten(2), vh.flatten(2)], -1).to(dev).bfloat16()
    q, k, v = (qkv[..., :h * 64].unflatten(-1, (h, 64)), qkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)),
               qkv[..., (h + 1) * 64:].unflatten(-1, (1, 64)))
    scale = 64 ** -0.5
    smax = float((torch.einsum("bihd,bjd->bhij", q.float()[:1, :256], k.float()[:1, :, 0]) * scale).abs().max())
    assert smax >= 40.0, smax
    slopes = torch.tensor([0.5, 0.25, 2.0 ** -4, 2.0 ** -6, 2.0 ** -8, 2.0 ** -10, 0.0, -0.01], device=dev)
    kmask = torch.ones(b, n, dtype=torch.bool, device=dev)
    kmask[0, 900:] = False
    kmask[1, 417:] = False
    kmask[1, 100:130] = False   # a hole: rows 100..129 have their own key masked
    d_o = torch.randn(b, n, h, 64, generator=g).to(dev).bfloat16()
    res = []
    try:
        for thr in (0.0, 30.0):
            ops.attn_set_band(thr)
            o, lse = ops.attn_fwd(q, k, v, kmask=kmask, slopes=slopes, causal=causal, scale=scale)
            dqkv = torch.zeros_like(qkv)
            dq, dk, dv = (dqkv[..., :h * 64].unflatten(-1, (h, 64)), dqkv[..., h * 64:(h + 1) * 64].unflatten(-1, (1, 64)),
                          dqkv[..., (h + 1) * 64:].unflatten(-1, (1, 64)))
            dsl = ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, kmask=kmask, slopes=slopes, causal=causal, scale=scale,
                               want_dslope=True)
            res.append((o.float(), lse, dqkv.float(), dsl))
    finally:
        ops.attn_set_band(30.0)
    (o0, l0, d0, s0), (o2, l2, d2, s2) = res
    assert torch.isfinite(o0).all() and torch.isfinite(d0).all() and torch.isfinite(s0).all()
    assert (o0 - o2).abs().max() <= 1e-5 * o0.abs().max() and (l0 - l2).abs().max() <= 1e-4
    assert (d0 - d2).abs().max() <= 1e-5 * d0.abs().max() and (s0 - s2).abs().max() <= 1e-4 * s0.abs().max()
    # heads with slope <= 0 are never skipped: bit-identical with the band on
    assert torch.equal(o0[:, :, 6:], o2[:, :, 6:])


def test_ffn_dropout_mask_is_consistent_and_unbiased(dev):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(5)
    T, I, p, seed = 512, 256, 0.1, 12345
    u = torch.randn(T, 2 * I, generator=g).to(dev).bfloat16()
    dout = torch.randn(T, I, generator=g).to(dev).bfloat16()
    out0 = ops.act_fwd(u, act=0, glu=True).float()
    out1 = ops.act_fwd(u, act=0, glu=True, p_drop=p, seed=seed).float()
    keep = out1 != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01
    assert rel_err(out1[keep], out0[keep] / (1 - p)) < 2e-2          # kept values scaled by 1/(1-p)
    assert (out1[~keep] == 0).all()
    again = ops.act_fwd(u, act=0, glu=True, p_drop=p, seed=seed).float()
    assert torch.equal(again, out1)                                  # same seed -> same mask
    other = ops.act_fwd(u, act=0, glu=True, p_drop=p, seed=seed + 1).float()
    assert (other != out1).float().mean().item() > 0.1
    du0 = ops.act_bwd(u, dout, act=0, glu=True).float()
    du1 = ops.act_bwd(u, dout, act=0, glu=True, p_drop=p, seed=seed).float()
    m2 = torch.cat([keep, keep], dim=1)
    assert rel_err(du1[m2], du0[m2] / (1 - p)) < 3e-2                # backward uses the same mask
    assert (du1[~m2] == 0).all()


@pytest.mark.parametrize("causal", [False, True])
def test_attention_dropout_forward_backward_share_one_mask(dev, causal):
    """The mask is a pure function of (seed, b, h, i, j): extract it with one-hot V, then check forward and all three
    backward kernels against an fp32 reference that uses exactly that mask."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(31 + int(causal))
    b, h, nq, nk, p, seed = 2, 2, 160, 192, 0.25, 777
    q = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()
    k = torch.randn(b, nk, 1, 64, generator=g).to(dev).bfloat16()
    v = torch.randn(b, nk, 1, 64, generator=g).to(dev).bfloat16()
    slopes = torch.tensor([0.3, 0.05], device=dev)
    scale = 64 ** -0.5
    # 1. reveal P_dropped column block by column block
    pd = torch.zeros(b, h, nq, nk, device=dev)
    for t in range(nk // 64):
        vi = torch.zeros(b, nk, 1, 64, device=dev)
        vi[:, 64 * t:64 * (t + 1), 0] = torch.eye(64, device=dev)
        o, _, _ = ops.attn_fwd(q, k, vi.bfloat16(), slopes=slopes, causal=causal, scale=scale, p_drop=p, seed=seed)
        pd[..., 64 * t:64 * (t + 1)] = o.float().permute(0, 2, 1, 3)
    qf, kf = q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3).expand(b, h, nk, 64)
    dist = torch.arange(nk, device=dev)[None, :] - (torch.arange(nq, device=dev)[:, None] + nk - nq)
    dots = qf @ kf.transpose(-1, -2) * scale - slopes.view(1, h, 1, 1) * dist.abs().float()
    if causal:
        dots = dots.masked_fill(dist > 0, -1.7014118e38)
    p0 = dots.softmax(-1)
    visible = p0 > 1e-3
    mask = (pd > 0)
    thr = p   # the drop rate is p itself (8-bit threshold + dithered 16-bit fraction), kept scores scaled by 1 / (1 - p)
    assert abs(mask[visible].float().mean().item() - (1 - thr)) < 0.02
    assert rel_err(pd[mask & visible], (p0 / (1 - thr))[mask & visible]) < 3e-2
    # 2. forward + backward with random V against the reference using the extracted mask
    mfull = torch.where(visible, mask, torch.ones_like(mask)).float()   # invisible entries: contribution ~ 0 either way
    qr, kr, vr = q.float().requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    dots_r = (qr.permute(0, 2, 1, 3) @ kr.permute(0, 2, 1, 3).expand(b, h, nk, 64).transpose(-1, -2)) * scale \
        - slopes.view(1, h, 1, 1) * dist.abs().float()
    if causal:
        dots_r = dots_r.masked_fill(dist > 0, -1.7014118e38)
    ref = ((dots_r.softmax(-1) * mfull / (1 - thr)) @ vr.permute(0, 2, 1, 3).expand(b, h, nk, 64)).permute(0, 2, 1, 3)
    d_o = torch.randn(b, nq, h, 64, generator=g).to(dev).bfloat16()
    ref.backward(d_o.float())
    o, lse, bits = ops.attn_fwd(q, k, v, slopes=slopes, causal=causal, scale=scale, p_drop=p, seed=seed)
    assert rel_err(o, ref) < 3e-2
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attn_bwd(q, k, v, o, d_o, lse, dq=dq, dk=dk, dv=dv, slopes=slopes, causal=causal, scale=scale, p_drop=p, dropbits=bits)
    assert rel_err(dq, qr.grad) < 4e-2
    assert rel_err(dk, kr.grad) < 4e-2
    assert rel_err(dv, vr.grad) < 4e-2


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.1, 0.25, 0.003])
def test_attention_dropout_rate_is_p(p):
    """F.dropout(p) drops with probability p (attend.py:122).  8.4 M keep bits of a full (unmasked, bidirectional, bias-free) launch:
    the observed rate is p within 5 standard errors -- for p = 0.1 that separates 0.1 from the 26/256 = 0.1016 a bare 8-bit
    threshold would give -- and every block of 32 x 64 scores sits at one of the two neighbouring 8-bit rates."""
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    b, h, n = 2, 4, 1024
    q = torch.randn(b, n, h, 64, generator=g).to(dev).bfloat16()
    kv = torch.randn(b, n, 1, 64, generator=g).to(dev).bfloat16()
    _, _, bits = ops.attn_fwd(q, kv, kv, p_drop=p, seed=4242)
    words = bits.view(torch.int16).to(torch.int32) & 0xffff
    kept = sum(((words >> i) & 1).sum().item() for i in range(16))
    total = words.numel() * 16
    assert total == b * h * n * n
    rate = 1.0 - kept / total
    se = (p * (1 - p) / total) ** 0.5
    assert abs(rate - p) < 5 * se + 2e-5, (rate, p, se)   # (+ the block-level dither's own sampling noise)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,ta,tb", [(37, 50, 27, False, False), (37, 50, 27, True, False), (37, 50, 27, False, True),
                                        (37, 50, 27, True, True), (300, 44, 260, False, False), (129, 12, 100, True, True)])
def test_gemm_accepts_widths_that_are_not_multiples_of_8(M, N, K, ta, tb):
    """Model widths off the 8-element grid (a `dim` of 100, a latent of 12): the wrapper pads the operands, results are unchanged."""
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M * 7 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).bfloat16().to(dev)
    b = torch.randn((K, N) if tb else (N, K), generator=g).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = ops.gemm(a, b, ta=ta, tb=tb, out_dtype=torch.float32, bias=bias)
    ref = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t()) + bias
    assert out.shape == (M, N)
    assert torch.allclose(out, ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    out16 = ops.gemm(a, b, ta=ta, tb=tb)
    assert torch.allclose(out16.float(), ref - bias, rtol=2e-2, atol=2e-2 * ref.abs().max().item())


def test_dec_head_sample_draws_from_the_top_k_softmax():
    """spn_dec_head_sample: banned ids never drawn, only the k largest logits drawn, frequencies = softmax(top-k logits / T)."""
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    D, widths, V = 96, [32, 64], [40, 150]
    tables = [torch.randn(v, w, generator=g).to(dev) * 0.35 for v, w in zip(V, widths)]
    e = torch.randn(D, generator=g).to(dev)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
    xn = torch.nn.functional.layer_norm(e, (D,), gamma, beta, 1e-5)
    col0, ks, T = [0, 32], [5, 12], 0.8
    want = []
    for t, c, w, k in zip(tables, col0, widths, ks):
        lg = t @ xn[c:c + w]
        lg[:2] = -float("inf")                                   # PAD / MASK banned (wrappers.py:368-369)
        top = torch.topk(lg, k)
        p = torch.zeros_like(lg)
        p[top.indices] = torch.softmax(top.values / T, -1)
        want.append(p.cpu())
    tokens = torch.zeros(2, 12, dtype=torch.int64, device=dev)
    pos = torch.zeros(1, dtype=torch.int32, device=dev)
    part = torch.zeros(16 * 8 * 2, device=dev)
    counter = torch.zeros(16, dtype=torch.int32, device=dev)
    logits = torch.zeros(16, 1024, device=dev)
    topk = torch.tensor(ks, dtype=torch.int32, device=dev)
    seed = torch.zeros(1, dtype=torch.int32, device=dev)
    N = 6000
    draws = torch.zeros(N, 2, dtype=torch.int64, device=dev)
    for i in range(N):
        tokens[1, 3] = 1
        tokens[1, 5] = 1
        seed.fill_(i * 7919 + 13)
        ops.dec_head_sample(tables, col0, [3, 5], D, e, gamma, beta, 1e-5, tokens, pos, part, counter, logits, topk, seed, temperature=T, slabs=8)
        draws[i, 0], draws[i, 1] = tokens[1, 3], tokens[1, 5]
    draws = draws.cpu()
    for j in range(2):
        freq = torch.bincount(draws[:, j], minlength=V[j]).float() / N
        assert float(freq[want[j] == 0].sum()) == 0.0            # nothing outside the top-k, nothing banned
        assert float((freq - want[j]).abs().max()) < 0.03, (freq, want[j])
    # a known token (no MASK in the cell) is left alone
    tokens[1, 3] = 17
    ops.dec_head_sample(tables, col0, [3, 5], D, e, gamma, beta, 1e-5, tokens, pos, part, counter, logits, topk, seed, temperature=T, slabs=8)
    assert int(tokens[1, 3]) == 17


@pytest.mark.parametrize("E,V,with_ln,pad_rows", [(128, [260, 132, 16, 92, 133, 40], True, True),     # one-hot MFMA scatter (recipe widths)
                                                  (128, [260, 132, 16, 92], False, False),           # ... without the LayerNorm
                                                  (32, [40, 36, 16, 28, 37], True, True)])           # LDS-atomic scatter (other widths)
def test_embedding_forward_backward_match_torch(E, V, with_ln, pad_rows):
    """spn_embed_fwd / spn_embed_bwd (gather + concat + LayerNorm and its backward into the per-key tables) against fp32 autograd.
    The E = 128 case runs the one-hot MFMA scatter, whose gradient tile is rounded to bf16 before it is accumulated in fp32."""
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5 + E)
    B, t = 3, 333                                     # not a multiple of the 64-row stage
    K, D = len(V), len(V) * E
    tables = [torch.randn(v, E, generator=g).to(dev) for v in V]
    tokens = torch.stack([torch.randint(0, v, (B, t), generator=g) for v in V], -1)
    if pad_rows:
        tokens[1, 200:] = 0                           # padding rows: token 0 = padding_idx, no table gradient
    tokens = tokens.to(dev)
    gamma = (torch.rand(D, generator=g) + 0.5).to(dev) if with_ln else None
    beta = (torch.randn(D, generator=g) * 0.1).to(dev) if with_ln else None
    y, mean, rstd = ops.embed_fwd(tables, tokens, gamma, beta)
    ref_tabs = [tb.clone().requires_grad_(True) for tb in tables]
    x = torch.cat([torch.nn.functional.embedding(tokens[..., k], ref_tabs[k], padding_idx=0) for k in range(K)], -1)
    # (nn.Embedding semantics: the padding row is looked up like any other; only its gradient is dropped)
    rg = gamma.clone().requires_grad_(True) if with_ln else None
    rb = beta.clone().requires_grad_(True) if with_ln else None
    yr = torch.nn.functional.layer_norm(x, (D,), rg, rb, 1e-5) if with_ln else x
    assert rel_err(y.view(B, t, D), yr) < 1e-2
    dy = torch.randn(B * t, D, generator=g).to(dev).bfloat16()
    yr.backward(dy.float().view(B, t, D))
    dgamma = torch.zeros(D, device=dev) if with_ln else None
    dbeta = torch.zeros(D, device=dev) if with_ln else None
    dts = ops.embed_bwd(tables, tokens, dy, gamma, mean, rstd, dgamma=dgamma, dbeta=dbeta, padding_idx=0)
    tol = 1.5e-2 if E == 128 else 2e-3
    for k in range(K):
        assert float(dts[k][0].abs().max()) == 0.0                       # padding row
        assert rel_err(dts[k], ref_tabs[k].grad) < tol, (k, rel_err(dts[k], ref_tabs[k].grad))
    if with_ln:
        assert rel_err(dgamma, rg.grad) < 2e-3 and rel_err(dbeta, rb.grad) < 2e-3


@pytest.mark.parametrize("persist", [2, 0])
@pytest.mark.parametrize("out_f32,tb", [(False, False), (True, False), (False, True)])
def test_gemm_persistent_tile_walk(dev, out_f32, tb, persist):
    """A launch of > 512 output tiles with ragged M and a bias against fp32 matmul, and written exactly once, as the one-block-per-CU
    tile walk (knobs gemm_persist / gemm_persist_bwd: next tile's loads in flight during the epilogue, store credits in the wait
    counts, edge tiles on the stronger wait; bf16 outputs only, fp32 outputs always take the plain grid) and as the plain grid."""
    from scoreperformer_amd import lib, ops
    old = lib.get_tuning("gemm_persist"), lib.get_tuning("gemm_persist_bwd")
    lib.set_tuning("gemm_persist", persist)
    lib.set_tuning("gemm_persist_bwd", 1 if persist else 0)
    try:
        _persistent_tile_walk_case(dev, out_f32, tb, ops)
    finally:
        lib.set_tuning("gemm_persist", old[0])
        lib.set_tuning("gemm_persist_bwd", old[1])


def _persistent_tile_walk_case(dev, out_f32, tb, ops):
    g = torch.Generator().manual_seed(11)
    M, N, K = 16384 + 8, 2304, 512
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    b = w.t().contiguous() if tb else w
    out = ops.gemm(a, b, tb=tb, out_dtype=torch.float32 if out_f32 else torch.bfloat16, bias=bias)
    ref = a.float() @ w.float().t() + bias
    assert rel_err(out, ref) < (2e-5 if out_f32 else 6e-3)
    # every row was written exactly once: a second call into a poisoned buffer gives the same result
    out2 = torch.full_like(out, float("nan"))
    ops.gemm(a, b, tb=tb, out=out2, bias=bias)
    assert torch.equal(out.view(torch.int32 if out_f32 else torch.int16), out2.view(torch.int32 if out_f32 else torch.int16))
