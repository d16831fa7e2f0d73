"""GPU: the elementwise / reduction / loss-side kernels against plain PyTorch fp32 at sizes that reach their fast paths
(unrolled column sums, activation backward with fused bias sums, run-length segment kernels, split MMD backward, masked casts)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize("glu,act,I", [(True, 0, 2048), (True, 0, 96), (False, 1, 512)])
def test_activation_forward_backward_and_fused_bias_sums(glu, act, I):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(1 + I)
    T = 2500
    W = 2 * I if glu else I
    u = torch.randn(T, W, generator=g).to(DEV).bfloat16()
    d = torch.randn(T, I, generator=g).to(DEV).bfloat16()
    fn = F.silu if act == 0 else F.gelu
    ur = u.float().requires_grad_(True)
    ref = ur[:, :I] * fn(ur[:, I:]) if glu else fn(ur)
    out = ops.act_fwd(u, act=act, glu=glu)
    assert rel_err(out, ref) < 1e-2
    ref.backward(d.float())
    bias_sum = torch.zeros(W, device=DEV) if ops.act_bwd_can_fuse_colsum(W, glu) else None
    du = ops.act_bwd(u, d, act=act, glu=glu, colsum=bias_sum)
    assert rel_err(du, ur.grad) < 1.5e-2
    if bias_sum is not None:   # the fused column sums are those of the bf16 du the consumer GEMMs read
        assert rel_err(bias_sum, du.float().sum(0)) < 1e-3


@pytest.mark.parametrize("N,dtype", [(512, torch.bfloat16), (640, torch.bfloat16), (36, torch.bfloat16), (512, torch.float32)])
def test_column_sums(N, dtype):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(N)
    x = torch.randn(5003, N, generator=g).to(DEV).to(dtype)       # odd row count: unrolled body + tail
    out = torch.full((N,), 2.0, device=DEV)
    ops.colsum(x, out=out)                                         # accumulates
    assert rel_err(out - 2.0, x.float().sum(0)) < 2e-5 * (1 if dtype == torch.float32 else 50)


def test_masked_cast_and_in_place_row_mask():
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 211, 512, generator=g).to(DEV)
    m = (torch.rand(3, 211, generator=g) < 0.7).to(DEV)
    y = ops.cast(x, torch.bfloat16, rowmask=m)
    assert torch.equal(y, (x * m[..., None]).bfloat16())
    z = x.bfloat16()
    ops.zero_masked_rows(z.view(-1, 512), m)
    assert torch.equal(z, y)
    assert torch.equal(ops.cast(y, torch.float32), y.float())


@pytest.mark.parametrize("d,S,mode", [(512, 90, "runs"), (32, 300, "runs"), (4, 2, "constant"), (20, 50, "unsorted")])
def test_segment_count_sum_gather(d, S, mode):
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(d + S)
    b, t = 5, 777
    if mode == "constant":
        seg = torch.ones(b, t, dtype=torch.long)
    elif mode == "runs":
        seg = torch.cumsum((torch.rand(b, t, generator=g) < (S - 2) / t).long(), 1).clamp_max(S - 1)
    else:
        seg = torch.randint(0, S, (b, t), generator=g)
    seg = seg.to(DEV)
    x = torch.randn(b, t, d, generator=g).to(DEV)
    mask = (torch.rand(b, t, generator=g) < 0.8).to(DEV)
    counts = ops.segment_count(seg, S)
    onehot = F.one_hot(seg, S).float()                               # [b, t, S]
    assert torch.equal(counts, onehot.sum(1))
    sums = ops.segment_sum(x, seg, S)
    want = torch.einsum("bts,btd->bsd", onehot, x)
    assert rel_err(sums, want) < 1e-5
    means = ops.segment_sum(x, seg, S, counts=counts, rowmask=mask)
    want_m = torch.einsum("bts,btd->bsd", onehot, x * mask[..., None]) / counts.clamp_min(1)[..., None]
    assert rel_err(means, want_m) < 1e-5
    # all levels in one pass (spn_segment_sum_multi / _gather_multi): this segmentation, a second one (sorted runs) and the sequence mean,
    # into column slices of wider per-level buffers; unsorted ids exercise the atomic path, whole runs the plain-store path
    seg2 = torch.cumsum((torch.rand(b, t, generator=g) < 0.3).long(), 1).to(DEV)
    S2 = int(seg2.max()) + 1
    seg3 = (~mask).long()
    levels = [(seg, S), (seg2, S2), (seg3, 2)]
    cnts = [ops.segment_count(sg_, k) for sg_, k in levels]
    if d % 4 == 0:
        bufs = [torch.zeros(b, k, d + 4 * i, device=DEV) for i, (_, k) in enumerate(levels)]
        for dt in (torch.float32, torch.bfloat16):
            for bf in bufs:
                bf.zero_()
            xin = x.to(dt)
            ops.segment_sum_multi(xin, mask, [l[0] for l in levels], cnts, [bf[..., :d] for bf in bufs], [l[1] for l in levels])
            for (sg_, k), cn, bf in zip(levels, cnts, bufs):
                oh = F.one_hot(sg_, k).float()
                want_l = torch.einsum("bts,btd->bsd", oh, xin.float() * mask[..., None]) / cn.clamp_min(1)[..., None]
                assert rel_err(bf[..., :d], want_l) < 1e-5
                assert float(bf[..., d:].abs().max()) == 0.0 if bf.shape[-1] > d else True
        srcs = [torch.randn(b, k, d + 4 * i, generator=g).to(DEV) for i, (_, k) in enumerate(levels)]
        got = ops.segment_gather_multi([s_[..., :d] for s_ in srcs], [l[0] for l in levels], cnts, [l[1] for l in levels], mask, d)
        want_g = sum(torch.gather(s_[..., :d] / cn.clamp_min(1)[..., None], 1, sg_[..., None].expand(b, t, d))
                     for s_, cn, (sg_, _) in zip(srcs, cnts, levels)) * mask[..., None]
        assert rel_err(got, want_g) < 1e-6
    src = torch.randn(b, S, d, generator=g).to(DEV)
    y = ops.segment_gather(src, seg, counts=counts, rowmask=mask)
    want_y = torch.gather(src / counts.clamp_min(1)[..., None], 1, seg[..., None].expand(b, t, d)) * mask[..., None]
    assert rel_err(y, want_y) < 1e-6
    wide = torch.randn(b, t, d + 8, generator=g).to(DEV)             # accumulate into a column slice of a wider buffer
    before = wide.clone()
    ops.segment_gather(src, seg, out=wide[..., 4:4 + d], accumulate=True)
    assert rel_err(wide[..., 4:4 + d], before[..., 4:4 + d] + torch.gather(src, 1, seg[..., None].expand(b, t, d))) < 1e-6
    assert torch.equal(wide[..., :4], before[..., :4]) and torch.equal(wide[..., 4 + d:], before[..., 4 + d:])
    # sums of a column slice of a wider buffer: 16-byte aligned offsets take the four-columns-per-thread kernel, others the scalar one
    for off in (4, 2):
        xs = wide[..., off:off + d]
        assert rel_err(ops.segment_sum(xs, seg, S, rowmask=mask), torch.einsum("bts,btd->bsd", onehot, xs * mask[..., None])) < 1e-5


@pytest.mark.parametrize("N,D", [(1500, 32), (300, 8), (2100, 4)])
def test_mmd_forward_backward(N, D):
    """MMDFn = compute_mmd of the reference (mmd_transformer.py:521-534) with 0/1 row weights; N >= 1024 takes the split backward."""
    from scoreperformer_amd import functional as F_
    g = torch.Generator().manual_seed(N)
    Z = 256
    y = torch.randn(N, D, generator=g).to(DEV)
    z = torch.randn(Z, D, generator=g).to(DEV)
    w = (torch.rand(N, generator=g) < 0.6).float().to(DEV)

    def kernel_mean(a, b_):
        d2 = (a[:, None, :] - b_[None, :, :]).pow(2).mean(2) / D
        return torch.exp(-d2)

    yr = y.clone().requires_grad_(True)
    ys = yr[w > 0]
    ref = kernel_mean(z, z).mean() + kernel_mean(ys, ys).mean() - 2 * kernel_mean(z, ys).mean()
    yt = y.clone().requires_grad_(True)
    got = F_.MMDFn.apply(yt, w, z)
    assert abs(float(got) - float(ref)) < 1e-5 + 1e-4 * abs(float(ref))
    ref.backward()
    got.backward()
    assert rel_err(yt.grad, yr.grad) < 1e-3
    assert float(yt.grad[w == 0].abs().max()) == 0.0


def test_mmd_loss_more_rows_than_the_cap_but_fewer_valid_ones_equals_the_oracle():
    """`MMDLoss.forward` (mmd_transformer.py:511-520): the reference gathers the VALID latents first and subsets only if MORE THAN 4096
    of them remain.  The product never gathers (no host sync): with > 4096 ROWS it keeps the 4096 largest random keys, invalid rows
    last -- so 6000 rows of which 3000 are valid must give exactly the oracle's value on the valid rows, forward and backward.
    (The benchmark configuration b=64, n=2048 takes this branch on three of four levels every step.)"""
    from oracle import ref_cpu
    from scoreperformer_amd.models.scoreperformer.mmd_transformer import MMDLoss
    g = torch.Generator().manual_seed(77)
    N, D = 6000, 8
    y = (0.8 * torch.randn(N, D, generator=g) + 0.1)
    z = torch.randn(256, D, generator=g)
    valid = torch.zeros(N, dtype=torch.bool)
    valid[torch.randperm(N, generator=g)[:3000]] = True
    yr = y.clone().requires_grad_(True)
    ref = ref_cpu.compute_mmd(z, yr[valid])
    ref.backward()
    crit = MMDLoss()
    assert N > crit.max_num_latents >= 3000
    for seed in (0, 1):                                   # whatever keys are drawn, the selection holds every valid row
        torch.manual_seed(seed)
        yt = y.to(DEV).view(3, 2000, D).clone().requires_grad_(True)
        got = crit(yt, mask=valid.to(DEV).view(3, 2000), z=z.to(DEV))
        assert abs(float(got) - float(ref)) < 1e-5 + 1e-4 * abs(float(ref)), (float(got), float(ref))
        got.backward()
        gt = yt.grad.view(N, D).cpu()
        assert rel_err(gt, yr.grad) < 1e-3
        assert float(gt[~valid].abs().max()) == 0.0


def test_mmd_loss_subset_of_more_than_the_cap_valid_rows_is_uniform_and_unbiased():
    """More than 4096 valid latents: the reference computes the loss on `latents[randperm(n)[:4096]]` (mmd_transformer.py:515-517), a
    uniform subset without replacement.  The product's top-k-of-random-keys selection must pick exactly 4096 rows, all of them valid,
    every valid row equally often, and the loss averaged over draws must agree with the exact expectation of the reference's estimator
    (computed with the oracle's kernel means on the full set: the diagonal of the y-y kernel weighs 1/4096 instead of 1/n)."""
    from oracle import ref_cpu
    from scoreperformer_amd.models.scoreperformer.mmd_transformer import MMDLoss
    g = torch.Generator().manual_seed(78)
    N, D, n_valid, cap, draws = 6500, 4, 6000, 4096, 100
    y = 0.7 * torch.randn(N, D, generator=g) + 0.3 * (torch.arange(N)[:, None] / N)     # a drift along the rows: position matters
    z = torch.randn(256, D, generator=g)
    valid = torch.zeros(N, dtype=torch.bool)
    valid[torch.randperm(N, generator=g)[:n_valid]] = True
    yv = y[valid]
    kyy = float(ref_cpu.gaussian_kernel_mean(yv, yv))
    off = (kyy * n_valid * n_valid - n_valid) / (n_valid * (n_valid - 1))               # mean over i != j
    expect = float(ref_cpu.gaussian_kernel_mean(z, z)) + (1.0 / cap + (cap - 1) / cap * off) \
        - 2 * float(ref_cpu.gaussian_kernel_mean(z, yv))
    crit = MMDLoss()
    assert crit.max_num_latents == cap
    yd, vd, zd = y.to(DEV), valid.to(DEV), z.to(DEV)
    torch.manual_seed(5)
    vals, picked = [], torch.zeros(N, device=DEV)
    for _ in range(draws):
        yt = yd.clone().requires_grad_(True)
        v = crit(yt, mask=vd, z=zd)
        v.backward()
        sel = yt.grad.abs().sum(-1) > 0
        assert int(sel.sum()) == cap and not bool((sel & ~vd).any())
        picked += sel.float()
        vals.append(float(v))
    vals = torch.tensor(vals, dtype=torch.float64)
    se = float(vals.std()) / draws ** 0.5
    assert abs(float(vals.mean()) - expect) < 4 * se + 1e-6, (float(vals.mean()), expect, se)
    assert float(vals.std()) > 0                                                        # the draws do differ
    freq = (picked / draws)[vd].cpu()
    p = cap / n_valid
    sd = (p * (1 - p) / draws) ** 0.5
    assert float((freq - p).abs().max()) < 5.5 * sd                                      # 6000 rows: max |z| ~ 4
    half = n_valid // 2
    assert abs(float(freq[:half].mean()) - float(freq[half:].mean())) < 5 * sd / half ** 0.5 * 2 ** 0.5 + 1e-3


def test_cross_entropy_forward_backward():
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(4)
    B, t, V = 3, 401, 165
    logits = (torch.randn(B * t, 168, generator=g) * 3).to(DEV)[:, :V]        # padded row stride
    labels = torch.randint(0, V, (B, t, 3), generator=g)
    labels[torch.rand(B, t, generator=g) < 0.3] = -100
    lab = labels.to(DEV)[..., 1]                                             # strided label column
    lse, sums, am = ops.ce_fwd(logits, V, lab, want_argmax=True)
    lr = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lr, lab.reshape(-1), ignore_index=-100, reduction="sum")
    assert abs(float(sums[0]) - float(ref)) < 1e-4 * float(ref) and float(sums[1]) == float((lab != -100).sum())
    assert torch.equal(am.long(), logits.argmax(-1)) and rel_err(lse, torch.logsumexp(logits, -1)) < 1e-6
    ref.backward()
    coef = torch.ones(1, device=DEV)
    dl = ops.ce_bwd(logits, V, lab, lse, coef)
    assert rel_err(dl[:, :V], lr.grad) < 1e-2 and float(dl[:, V:].abs().max()) == 0.0


@pytest.mark.parametrize("M,I,K,act,p", [(256, 128, 256, 0, 0.0), (1000, 256, 512, 0, 0.1), (4096, 2048, 512, 1, 0.0),
                                         (8192 + 8, 2048, 512, 0, 0.1), (131072, 2048, 512, 0, 0.1)])
@pytest.mark.parametrize("persist", [2, 0])
def test_gemm_glu_equals_gemm_then_activation(M, I, K, act, p, persist):
    """Fused GLU projection: u is the plain GEMM's output and g the stand-alone activation kernel's output on that u, bit for bit
    (same rounding point, same dropout mask), at edge row counts and at the benchmark's FFN shape; as one persistent block per CU
    (the default from two rounds of tiles on) and as one block per tile."""
    from scoreperformer_amd import lib, ops
    old = lib.get_tuning("glu_persist")
    lib.set_tuning("glu_persist", persist)
    try:
        _glu_case(M, I, K, act, p, ops)
    finally:
        lib.set_tuning("glu_persist", old)


def _glu_case(M, I, K, act, p, ops):
    gen = torch.Generator(device="cuda").manual_seed(M + I)
    x = (torch.randn(M, K, device="cuda", generator=gen) * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda", generator=gen) * K ** -0.5).bfloat16()
    b = torch.randn(2 * I, device="cuda", generator=gen) * 0.1
    assert ops.gemm_glu_ok(M, I, K)
    u, g = ops.gemm_glu(x, w, b, act=act, p_drop=p, seed=1234)
    u_ref = ops.gemm(x, w, tb=False, out_dtype=torch.bfloat16, bias=b)
    assert torch.equal(u.view(torch.int16), u_ref.view(torch.int16))
    g_ref = ops.act_fwd(u_ref, act=act, glu=True, p_drop=p, seed=1234)
    assert torch.equal(g.view(torch.int16), g_ref.view(torch.int16))
    if p > 0:
        kept = (g != 0).float().mean().item()
        assert abs(kept - (1 - p)) < 0.01


@pytest.mark.parametrize("M,I,K,act,p", [(256, 256, 256, 0, 0.0), (904, 512, 512, 0, 0.1), (4096, 2048, 512, 1, 0.1),
                                         (8192 + 8, 2048, 512, 0, 0.1), (131008, 2048, 512, 0, 0.1)])
@pytest.mark.parametrize("duo", [2, 1, 0])
def test_gemm_glu_bwd_equals_gemm_then_activation_backward(M, I, K, act, p, duo):
    """Gated backward epilogue: du and the bias column sums equal the input-gradient GEMM followed by the stand-alone activation
    backward, bit for bit (dg rounded to bf16 at the same point, same dropout mask), at edge row counts and at the benchmark's shape;
    on all three tile kernels (two 8-wave workgroups per CU = the default, two 4-wave workgroups, the ping-pong kernel)."""
    from scoreperformer_amd import lib, ops
    old = lib.get_tuning("glu_bwd_duo")
    lib.set_tuning("glu_bwd_duo", duo)
    try:
        _glu_bwd_case(M, I, K, act, p, ops)
    finally:
        lib.set_tuning("glu_bwd_duo", old)


def _glu_bwd_case(M, I, K, act, p, ops):
    gen = torch.Generator(device="cuda").manual_seed(M + I + 1)
    dy = (torch.randn(M, K, device="cuda", generator=gen) * 0.5).bfloat16()
    w2 = (torch.randn(K, I, device="cuda", generator=gen) * K ** -0.5).bfloat16()
    u = torch.randn(M, 2 * I, device="cuda", generator=gen).bfloat16()
    assert ops.gemm_glu_bwd_ok(M, I, K)
    cs = torch.zeros(2 * I, device="cuda")
    du = ops.gemm_glu_bwd(dy, w2, u, act=act, p_drop=p, seed=4321, colsum=cs)
    dg = ops.gemm(dy, w2, tb=True, out_dtype=torch.bfloat16)
    cs_ref = torch.zeros(2 * I, device="cuda")
    du_ref = ops.act_bwd(u, dg, act=act, glu=True, p_drop=p, seed=4321, colsum=cs_ref)
    assert torch.equal(du.view(torch.int16), du_ref.view(torch.int16))
    exact = du_ref.float().sum(0)
    tol = 1e-3 * float(du_ref.float().abs().sum(0).max()) + 1e-3
    assert float((cs - exact).abs().max()) < tol and float((cs_ref - exact).abs().max()) < tol
    du2 = ops.gemm_glu_bwd(dy, w2, u, act=act, p_drop=p, seed=4321)   # without the column sums
    assert torch.equal(du2.view(torch.int16), du.view(torch.int16))


@pytest.mark.parametrize("duo", [0, 1, 2])
@pytest.mark.parametrize("M", [8200, 9000, 256 + 128, 256 + 8, 512 - 8])
def test_gemm_glu_bwd_partial_rows_stay_inside_the_buffer(M, duo):
    """The column-sum partial buffer is [ceil(M / 128), 2I] (include/spn.h).  With 0 < M % 256 <= 128 the last 256-row tile's second
    128-row wave slab starts at or beyond M and owns no partial row: a canary row behind the buffer must survive (both tile kernels)."""
    import ctypes
    from ctypes import c_float, c_int
    from scoreperformer_amd import lib, ops
    I, K = 256, 256
    gen = torch.Generator(device="cuda").manual_seed(M)
    dy = (torch.randn(M, K, device="cuda", generator=gen) * 0.5).bfloat16()
    w2 = (torch.randn(K, I, device="cuda", generator=gen) * K ** -0.5).bfloat16()
    u = torch.randn(M, 2 * I, device="cuda", generator=gen).bfloat16()
    rows = (M + 127) // 128
    part = torch.full((rows + 1, 2 * I), 12345.0, device="cuda")
    du = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16)
    old = lib.get_tuning("glu_bwd_duo")
    lib.set_tuning("glu_bwd_duo", duo)
    try:
        ops.call("spn_gemm_glu_bwd", ops.ptr(dy), ops.ptr(w2), ops.ptr(u), ops.ptr(du), ops.ptr(part), c_int(M), c_int(I), c_int(K),
                 c_int(K), c_int(I), c_int(2 * I), c_int(2 * I), c_int(0), c_float(0.0), ctypes.c_uint(0), ops.stream_ptr())
    finally:
        lib.set_tuning("glu_bwd_duo", old)
    torch.cuda.synchronize()
    assert bool((part[rows] == 12345.0).all()), "partial row past ceil(M / 128) was written"
    ref = ops.act_bwd(u, ops.gemm(dy, w2, tb=True, out_dtype=torch.bfloat16), act=0, glu=True)
    assert torch.equal(du.view(torch.int16), ref.view(torch.int16))
    exact = ref.float().sum(0)
    assert float((part[:rows].sum(0) - exact).abs().max()) < 1e-3 * float(ref.float().abs().sum(0).max()) + 1e-3


def test_gemm_glu_rejects_unsupported_shapes():
    from scoreperformer_amd import ops
    assert not ops.gemm_glu_ok(64, 128, 256) and not ops.gemm_glu_ok(256, 192, 256) and not ops.gemm_glu_ok(256, 128, 200)
    x = torch.zeros(256, 200, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(256, 200, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ops.SpnError):
        ops.gemm_glu(x, w, None, act=0)


def test_feedforward_fused_glu_matches_two_kernel_path(monkeypatch):
    """FeedForward(glu) forward and every gradient are identical with the fused projection and with GEMM + activation kernels."""
    import scoreperformer_amd.functional as F_
    from scoreperformer_amd.modules.transformer.feedforward import FeedForward
    torch.manual_seed(0)
    ff = FeedForward(dim=512, mult=4, glu=True, swish=True, dropout=0.1, no_bias=False).cuda().train()
    x0 = torch.randn(4, 256, 512, device="cuda")
    outs = []
    for fuse, ffn in ((True, True), (True, False), (False, False)):   # whole-block node / fused projection only / separate kernels
        monkeypatch.setattr(F_, "GLU_FUSE", fuse)
        monkeypatch.setattr(F_, "FFN_FUSE", ffn)
        F_._seed_state["base"] = None   # same dropout seeds in both runs
        ff.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = ff(x.bfloat16())
        y.float().square().mean().backward()
        outs.append((y.detach().float(), x.grad.clone(), [p.grad.clone() for p in ff.parameters()]))
    (y1, dx1, g1) = outs[0]
    for (y2, dx2, g2) in outs[1:]:
        assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
        for a, b in zip(g1, g2):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)   # bias column sums: same values, another summation order


def test_gemm_random_shapes_full_tensor_screen():
    """Race / edge screen of the ping-pong GEMM (peeled K loop, asm LDS DMA, pre-read fragments, split-K through a workspace): random
    shapes in every layout, compared in full with torch's fp32 matmul of the same bf16 operands (tools/stress_gemm.py, 80 cases)."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stress_gemm", os.path.join(root, "tools", "stress_gemm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(80, 7, verbose=False) == 0


@pytest.mark.parametrize("K", [64, 96, 160, 224])
@pytest.mark.parametrize("tb,f32,extras", [(False, False, True), (True, True, True), (False, True, False)])
def test_gemm_short_contraction_on_the_duo_kernel(K, tb, f32, extras):
    """K < 256 (below the ping-pong kernel's range: the AdaLN condition projections have K = 64) goes to the two-workgroups-per-CU
    kernel when the launch has >= 512 tiles: two to seven K tiles of 32, ragged M, bias / residual, against fp32 matmul in full."""
    from scoreperformer_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(K + 7 * tb + 13 * f32)
    M, N = 256 * 70 + 8, 1024          # 71 x 8 = 568 tiles of 256 x 128
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    b = w.t().contiguous() if tb else w
    bias = torch.randn(N, device=dev, generator=g) if extras else None
    res = torch.randn(M, N, device=dev, generator=g) if (extras and f32) else None
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    ops.gemm(a, b, tb=tb, out=out, bias=bias, residual=res)
    ref = a.float() @ w.float().t()
    if bias is not None:
        ref = ref + bias
    if res is not None:
        ref = ref + res
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (2e-5 * K ** 0.5 if f32 else 1e-2) and bool(torch.isfinite(out.float()).all()), err


@pytest.mark.parametrize("T", [77, 1024, 8200])
def test_adaln_forward_with_the_projection_inside_the_kernel(T):
    """AdaptiveLayerNorm (modules/layers.py:31-47) with gamma | beta = Linear(condition) computed on the matrix cores inside the LayerNorm
    forward (D = 512, C = 64): y, mean, rstd and the bf16 gamma rows against fp32 torch, at ragged token counts; the gamma rows are what
    spn_layernorm_bwd_gb16 reads in the backward, checked through the autograd function against the unfused path."""
    from scoreperformer_amd import ops
    import scoreperformer_amd.functional as F_
    D, C = 512, 64
    assert ops.adaln_ok(D, C) and not ops.adaln_ok(128, 32)
    g = torch.Generator(device="cuda").manual_seed(T)
    x = torch.randn(T, D, device="cuda", generator=g) * 2 + 0.3
    cond = torch.randn(T, C, device="cuda", generator=g).bfloat16()
    w = (torch.randn(2 * D, C, device="cuda", generator=g) * 0.1)
    bias = torch.cat([torch.ones(D), torch.zeros(D)]).cuda() + 0.05 * torch.randn(2 * D, device="cuda", generator=g)
    y, mean, rstd, gam = ops.adaln_fwd(x, cond, w.bfloat16(), bias)
    gb = cond.float() @ w.bfloat16().float().t() + bias
    ref = gb[:, :D] * F.layer_norm(x, (D,)) + gb[:, D:]
    assert rel_err(y, ref) < 6e-3 and rel_err(gam, gb[:, :D]) < 5e-3        # bf16 outputs of fp32 values
    assert rel_err(mean, x.mean(1)) < 1e-5 and rel_err(rstd, (x.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    # through autograd: fused forward + wave-per-row backward on the gamma rows == GEMM + LayerNorm path
    dy = torch.randn(T, D, device="cuda", generator=g)
    outs = []
    for fused in (True, False):
        F_.ADALN_FUSED = fused
        try:
            xs, cs = x.clone().requires_grad_(True), cond.float().requires_grad_(True)
            ws, bs = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
            yy = F_.ada_layer_norm(xs.view(1, T, D), cs.view(1, T, C), ws, bs)
            yy.float().backward(dy.view(1, T, D))
            outs.append((yy.detach().float(), xs.grad, cs.grad, ws.grad, bs.grad))
        finally:
            F_.ADALN_FUSED = True
    for a, b_ in zip(*outs):
        assert rel_err(a, b_) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_standalone_dropout_forward_backward(dtype):
    """ops.dropout / F_.dropout (spn_dropout: `emb_dropout`, the Dropout behind a post-activation LayerNorm): kept entries scaled by
    1 / (1 - p), the drop rate is p, the backward applies the SAME mask to the gradient, p = 0 / eval are the identity."""
    from scoreperformer_amd import functional as F_, ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4, 1000, 512, generator=g) + 3.0).to(DEV).to(dtype)       # no zeros in x: dropped <=> y == 0
    p = 0.2
    y = ops.dropout(x, p, seed=77)
    kept = y != 0
    rate = 1.0 - float(kept.float().mean())
    assert abs(rate - p) < 5 * (p * (1 - p) / x.numel()) ** 0.5 + 1e-4, rate
    torch.testing.assert_close(y[kept].float(), (x[kept].float() / (1 - p)).to(dtype).float(), rtol=2e-2 if dtype == torch.bfloat16 else 2e-5, atol=0)   # (p is quantised to 16 bits: 1 / (1 - p) within 4e-6)
    assert torch.equal(ops.dropout(x, p, seed=77), y) and not torch.equal(ops.dropout(x, p, seed=78) != 0, kept)
    # rows and columns are not correlated: the keep rate per row and per column stays within 6 sigma
    assert float((kept.float().mean(-1) - (1 - p)).abs().max()) < 6 * (p * (1 - p) / 512) ** 0.5
    assert float((kept.view(-1, 512).float().mean(0) - (1 - p)).abs().max()) < 6 * (p * (1 - p) / 4000) ** 0.5
    xr = x.clone().requires_grad_(True)
    out = F_.dropout(xr, p, training=True)
    out.backward(torch.ones_like(out))
    assert torch.equal(xr.grad != 0, out != 0)
    torch.testing.assert_close(xr.grad[out != 0].float(), torch.full_like(xr.grad[out != 0].float(), 1 / (1 - p)), rtol=1e-2, atol=0)
    assert F_.dropout(xr, p, training=False) is xr and F_.dropout(xr, 0.0) is xr


@pytest.mark.parametrize("T,D", [(4099, 512), (1000, 256), (777, 384)])
def test_adaptive_layernorm_backward_leaves_the_bias_column_sums(T, D):
    """spn_layernorm_bwd_gb16_colsum: the column sums of the (dy * xhat | dy) rows -- the bias gradient of the condition Linear
    (modules/layers.py:38) -- accumulate into a [2D] target in the LayerNorm backward's own pass (branch-free kernel at D = 256 / 512,
    the general kernel elsewhere); dx and the dgb rows are those of the call without the target."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(T + D)
    x = torch.randn(T, D, generator=g).to(DEV)
    gb = (torch.randn(T, 2 * D, generator=g) * 0.5 + 1.0).to(DEV).bfloat16()
    dy = torch.randn(T, D, generator=g).to(DEV).bfloat16()
    dres = torch.randn(T, D, generator=g).to(DEV)
    _, mean, rstd = ops.layernorm_fwd(x, None, None, gb)
    dx0, dgb0 = ops.layernorm_bwd(x, dy, None, gb, mean, rstd, dres=dres, want_dgb=True, want_dx16=True)
    target = torch.full((2 * D,), 3.0, device=DEV)
    dx1, dgb1 = ops.layernorm_bwd(x, dy, None, gb, mean, rstd, dres=dres, want_dgb=True, want_dx16=True, dgb_colsum=target)
    assert torch.equal(dx0, dx1) and torch.equal(dgb0.view(torch.int16), dgb1.view(torch.int16))
    assert torch.equal(dx1._spn_bf16.view(torch.int16), dx0._spn_bf16.view(torch.int16))
    xh = (x - mean[:, None]) * rstd[:, None]
    exact = torch.cat([(dy.float() * xh).sum(0), dy.float().sum(0)])
    assert rel_err(target - 3.0, exact) < 2e-4


@pytest.mark.parametrize("M,N,K,ta,tb", [(5000, 4, 572, False, False), (5000, 20, 544, False, False), (300, 32, 512, False, False),
                                        (5000, 572, 8, False, True), (8, 572, 5000, True, True), (32, 544, 3000, True, True),
                                        (130, 70, 200, False, False), (4, 16, 70000, True, True)])
def test_exact_fp32_gemm_all_tile_shapes(M, N, K, ta, tb):
    """spn_gemm_f32 (VAE heads `MMDVAE.linear`, mmd_transformer.py:53-56, and their backward contractions): the 64x64 tile and the two
    skinny tiles (N <= 32: 64x16; M <= 32: 16x64), with bias, row mask, accumulation and the split-K weight-gradient path, against torch
    fp32 -- exact up to summation order."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    b = torch.randn((K, N) if tb else (N, K), generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    A = a.t() if ta else a
    B = b if tb else b.t()
    ref = A.double() @ B.double()
    tol = 2e-6 * float(ref.abs().max()) * max(1.0, (K / 512) ** 0.5)
    out = ops.gemm_f32(a, b, ta=ta, tb=tb)
    assert float((out.double() - ref).abs().max()) < tol
    out = ops.gemm_f32(a, b, ta=ta, tb=tb, bias=bias, alpha=0.5)
    assert float((out.double() - (0.5 * ref + bias.double())).abs().max()) < tol
    base = torch.randn(M, N, generator=g).to(DEV)
    acc = base.clone()
    ops.gemm_f32(a, b, ta=ta, tb=tb, out=acc, accumulate=True)
    assert float((acc.double() - (base.double() + ref)).abs().max()) < tol
    if not ta:
        mask = (torch.rand(M, generator=g) < 0.7).to(DEV)
        out = ops.gemm_f32(a, b, ta=ta, tb=tb, bias=bias, rowmask=mask)
        assert float((out.double() - (ref + bias.double()) * mask[:, None].double()).abs().max()) < tol


def test_gradient_norm_has_a_fixed_summation_order():
    """`spn_sumsq_det` (the clip's gradient norm): the same bits for the same gradient on every launch -- data-parallel replicas hold the
    same all-reduced gradient and must get the same clip coefficient, or they drift apart an ulp per step (found by the first two-process
    run, tests/test_dp_gpu.py).  72 M elements, 40 launches: one distinct result, equal to the fp64 sum to 1e-6; the float-atomic
    version is checked to the same accuracy only (its bits do vary)."""
    from scoreperformer_amd import ops
    g = torch.randn(71895400, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) * 1e-2
    ws = torch.empty(ops.sumsq_ws_floats(), device="cuda")
    got = set()
    for _ in range(40):
        out = torch.zeros(1, device="cuda")
        ops.sumsq(g, out=out, ws=ws)
        got.add(out.view(torch.int32).item())
    assert len(got) == 1, got
    want = float(g.double().pow(2).sum())
    val = torch.tensor([got.pop()], dtype=torch.int32).view(torch.float32).item()
    assert abs(val - want) <= 1e-6 * want
    out = torch.ones(1, device="cuda")
    ops.sumsq(g, out=out, ws=ws)                      # accumulates onto what is there
    assert abs(out.item() - (1.0 + want)) <= 1e-6 * want
    out = torch.zeros(1, device="cuda")
    ops.sumsq(g[: 1000003], out=out)                  # atomic version, ragged length
    assert abs(out.item() - float(g[: 1000003].double().pow(2).sum())) <= 1e-5 * float(g[: 1000003].double().pow(2).sum())
    out = torch.zeros(1, device="cuda")
    ops.sumsq(g[: 1000000], out=out, ws=ws)           # fixed-order version on a short, 16-byte aligned slice
    assert abs(out.item() - float(g[: 1000000].double().pow(2).sum())) <= 1e-6 * float(g[: 1000000].double().pow(2).sum())


@pytest.mark.parametrize("M,N,K", [(512, 4096, 512), (280, 640, 512), (333, 512, 2048), (48, 64, 16), (130, 200, 572), (1, 64, 64)])
def test_exact_fp32_gemm_on_the_fp32_matrix_cores(dev, M, N, K):
    """spn_gemm_f32 with both operands contiguous along K (x [M, K], nn.Linear weight [N, K]) runs on v_mfma_f32_32x32x2_f32 tiles
    (csrc/gemm_f32.hip gemm_f32_mfma_kernel; M >= 48, N >= 64, K in whole float4): fp32 products and sums -- against the fp64 product at
    fp32 rounding, with bias, row mask, alpha and accumulation, tile edges in M, N and K, and against the VALU tile kernel (knob off)."""
    from scoreperformer_amd import lib, ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) > 0.2).to(dev)
    c0 = torch.randn(M, N, generator=g).to(dev)
    want = (0.5 * (a.double() @ w.double().t()) + bias.double()) * mask.double()[:, None] + c0.double()
    before = lib.get_tuning("gemm_f32_mfma")
    try:
        outs = []
        for knob in (1, 0):
            lib.set_tuning("gemm_f32_mfma", knob)
            out = c0.clone()
            ops.gemm_f32(a, w, bias=bias, rowmask=mask, out=out, alpha=0.5, accumulate=True)
            outs.append(out)
            assert (out.double() - want).abs().max() <= 2e-6 * K ** 0.5 + 1e-6 * want.abs().max()
        # the two kernels round differently (summation order), by no more than fp32 accumulation allows
        assert (outs[0] - outs[1]).abs().max() <= 4e-6 * max(1.0, float(want.abs().max()))
        # a strided view of A (rows of a wider buffer) and a plain product without the optional operands
        wide = torch.randn(M, K + 8, generator=g).to(dev)
        lib.set_tuning("gemm_f32_mfma", 1)
        got = ops.gemm_f32(wide[:, :K], w)
        assert (got.double() - wide[:, :K].double() @ w.double().t()).abs().max() <= 2e-6 * K ** 0.5
    finally:
        lib.set_tuning("gemm_f32_mfma", before)


@pytest.mark.parametrize("n,t0,h,kvh,with_slopes", [(500, 0, 8, 1, True), (37, 300, 8, 1, True), (260, 5, 4, 4, True), (64, 0, 2, 1, False)])
def test_window_attention_rows_against_fp64(dev, n, t0, h, kvh, with_slopes):
    """spn_dec_attn_rows (the attention of a render window's batched re-priming: query rows t0 .. t0 + n - 1, each over the cached keys
    j <= t, ALiBi distance t - j, multi-query or one K/V head per head) against an fp64 softmax: the batched form of the decode step
    (8 keys per lane group scored together, two-level merge of the 32 groups) at fp32 accuracy."""
    from scoreperformer_amd import ops
    g = torch.Generator().manual_seed(n + 31 * t0)
    L = t0 + n
    q = torch.randn(n, (h + 2 * kvh) * 64, generator=g).to(dev)            # rows of a fused q | k | v buffer: only the q part is read
    kc = torch.randn(L, kvh * 64, generator=g).to(dev)
    vc = torch.randn(L, kvh * 64, generator=g).to(dev)
    slopes = (2.0 ** -torch.arange(1, h + 1).float()).to(dev) if with_slopes else None
    out = torch.zeros(n, h * 64, device=dev)
    ops.dec_attn_rows(q, kc, vc, slopes, t0, out, h=h, kvh=kvh, scale=0.125)
    qd = q[:, :h * 64].double().view(n, h, 64)
    kd, vd = kc.double().view(L, kvh, 64), vc.double().view(L, kvh, 64)
    if kvh == 1:
        kd, vd = kd.expand(L, h, 64), vd.expand(L, h, 64)
    s = torch.einsum("rhd,jhd->hrj", qd, kd) * 0.125
    t = torch.arange(t0, t0 + n, device=dev)[:, None]
    j = torch.arange(L, device=dev)[None, :]
    if with_slopes:
        s = s - slopes.double()[:, None, None] * (t - j).clamp_min(0).double()[None]
    s = s.masked_fill((j > t)[None], float("-inf"))
    want = torch.einsum("hrj,jhd->rhd", s.softmax(-1), vd).reshape(n, h * 64)
    assert (out.double() - want).abs().max() <= 2e-5
