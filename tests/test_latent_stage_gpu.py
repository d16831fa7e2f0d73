"""GPU: the fused latent stage (csrc/latent.hip) against the tensor-op formulation of the same reference lines.

* `spn_latent_select` -- `latents[mask]` + `randperm(N)[:max_num_latents]` (mmd_transformer.py:511-517): exact set properties (only valid
  latents, exactly min(#valid, K), no duplicates, rows copied bit for bit), uniformity over seeds, and the deadpan sums (mmd:232-237).
* `functional.latent_losses` -- value and gradient against `MMDFn` over ALL valid latents (the subset is the whole set when it fits) and
  the tensor-op deadpan loss through torch autograd.
* `spn_latent_drop` -- explicit masks against the tensor-op scatter / OR / concat (mmd:249-253,275-283), bit for bit.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _level(dev, b, S, D, p_valid, seed):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(b, S, D, generator=g)
    valid = torch.rand(b, S, generator=g) < p_valid
    return lat.to(dev), valid.to(dev)


@pytest.mark.parametrize("b,S,D,K,p_valid", [(4, 40, 8, 4096, 0.7), (64, 1082, 4, 4096, 0.9), (64, 155, 20, 4096, 1.0), (3, 1, 32, 4096, 1.0),
                                            (16, 600, 32, 512, 0.05), (2, 2100, 16, 4096, 0.0)])
def test_select_picks_a_subset_of_the_valid_latents(dev, b, S, D, K, p_valid):
    from scoreperformer_amd import ops
    lat, valid = _level(dev, b, S, D, p_valid, seed=b * S + D)
    N = b * S
    K = min(K, N)
    deadpan = (torch.arange(b, device=dev) % 3 == 0)
    y, w, slot, dead = ops.latent_select(lat, valid, deadpan, K, seed=12345)
    nvalid = int(valid.sum())
    total = min(nvalid, K)
    assert int(w.sum()) == total and bool((w[:total] == 1).all()) and bool((w[total:] == 0).all())
    sel = slot >= 0
    assert int(sel.sum()) == total
    assert not bool(sel[~valid.reshape(-1)].any())                                 # only valid latents
    picked = slot[sel].long()
    assert sorted(picked.tolist()) == list(range(total))                           # every row of y is claimed once
    flat = lat.reshape(N, D)
    assert torch.equal(y[picked], flat[sel])                                       # rows copied bit for bit
    assert float(y[total:].abs().max() if total < K else 0.0) == 0.0
    # deadpan sums of the level
    wd = (deadpan[:, None] & valid).float()
    sq = lat * lat * wd[..., None]
    assert abs(float(dead[0]) - float(sq.sum())) <= 1e-4 * max(1.0, float(sq.sum()))
    assert float(dead[1]) == float(wd.sum()) and float(dead[2]) == float((sq != 0).any())
    # the backward scatter: unselect(select) puts every picked row back, zeros elsewhere
    dy = torch.randn(K, D, device=dev)
    back = ops.latent_unselect(dy, slot, lat, valid, None, dead, None).reshape(N, D)
    assert torch.equal(back[sel], dy[picked]) and float(back[~sel].abs().max() if int((~sel).sum()) else 0.0) == 0.0


def test_select_is_uniform_over_the_valid_latents(dev):
    """Every valid latent is picked with probability K / #valid: 400 seeds on 1000 valid latents of 1200, K = 250; the per-latent
    frequency is binomial(400, 0.25) -- mean 100, sd 8.7: all within 6 sd, their mean within 1 %."""
    from scoreperformer_amd import ops
    b, S, D, K = 4, 300, 4, 250
    lat = torch.randn(b, S, D, device=dev)
    valid = torch.ones(b, S, dtype=torch.bool, device=dev)
    valid.view(-1)[::6] = False
    counts = torch.zeros(b * S, device=dev)
    sets = set()
    for seed in range(400):
        _, _, slot, _ = ops.latent_select(lat, valid, None, K, seed=seed * 7919 + 1)
        counts += (slot >= 0).float()
        sets.add(tuple(torch.nonzero(slot >= 0).flatten()[:8].tolist()))
    assert float(counts[~valid.view(-1)].sum()) == 0
    c = counts[valid.view(-1)]
    assert abs(float(c.mean()) - 100.0) < 1.0 and float(c.min()) > 100 - 6 * 8.7 and float(c.max()) < 100 + 6 * 8.7
    assert len(sets) > 390                                                          # seeds give different subsets


@pytest.mark.parametrize("with_deadpan", [False, True])
def test_latent_losses_match_the_tensor_op_formulation(dev, with_deadpan):
    """N <= K: the subset is every valid latent, so the value must equal compute_mmd over `latents[mask]` (MMDFn with 0/1 weights) and
    the deadpan MSE, and the gradient must equal theirs (fp32 summation order differs: 2e-5 relative)."""
    from scoreperformer_amd import functional as F_
    b, S, D = 8, 90, 20
    lat, valid = _level(dev, b, S, D, 0.8, seed=3)
    deadpan = (torch.arange(b, device=dev) % 2 == 0) if with_deadpan else None
    z = torch.randn(256, D, device=dev)
    a = lat.clone().requires_grad_(True)
    mmd, dead, flag = F_.latent_losses(a, valid, deadpan, z, max_num_latents=4096, weight=0.5)
    (mmd * 3.0 + dead * 2.0).backward()
    r = lat.clone().requires_grad_(True)
    want_mmd = 0.5 * F_.MMDFn.apply(r.reshape(-1, D), valid.reshape(-1).float(), z)
    if with_deadpan:
        wd = (deadpan[:, None] & valid).float()
        sq = r * r * wd[..., None]
        want_dead = sq.sum() / (wd.sum() * D).clamp_min(1.0)
        want_flag = float((sq != 0).any())
    else:
        want_dead, want_flag = r.sum() * 0.0, 0.0
    (want_mmd * 3.0 + want_dead * 2.0).backward()
    assert abs(float(mmd) - float(want_mmd)) <= 2e-5 * max(1.0, abs(float(want_mmd)))
    assert abs(float(dead) - float(want_dead)) <= 2e-5 * max(1.0, abs(float(want_dead))) and float(flag) == want_flag
    assert float((a.grad - r.grad).norm()) <= 2e-5 * float(r.grad.norm())
    assert float(a.grad[~valid].abs().max()) == 0.0


def test_latent_losses_subsample_large_levels(dev):
    """More than max_num_latents valid latents: the loss is the MMD of SOME 4096 of them -- statistically the same number (latents drawn
    from one distribution), gradient only on 4096 rows, different rows for different calls."""
    from scoreperformer_amd import functional as F_
    b, S, D = 64, 300, 8
    lat, valid = _level(dev, b, S, D, 0.95, seed=5)
    z = torch.randn(256, D, device=dev)
    vals, rows = [], []
    for _ in range(3):
        a = lat.clone().requires_grad_(True)
        mmd, _, _ = F_.latent_losses(a, valid, None, z, max_num_latents=4096)
        mmd.backward()
        touched = a.grad.abs().sum(-1) != 0
        assert int(touched.sum()) == 4096 and not bool(touched[~valid].any())
        vals.append(float(mmd)); rows.append(touched)
    assert max(vals) - min(vals) < 0.05 * max(abs(v) for v in vals) + 1e-3
    assert not torch.equal(rows[0], rows[1])


@pytest.mark.parametrize("inclusive", [False, True])
def test_latent_drop_with_given_masks_equals_the_tensor_ops(dev, inclusive):
    from scoreperformer_amd import functional as F_
    b, n = 5, 70
    Ls = [6, 5, 3, 2]
    g = torch.Generator().manual_seed(11)
    mask = (torch.arange(n)[None] < torch.randint(n // 2, n + 1, (b, 1), generator=g)).to(dev)
    deadpan = torch.tensor([False, True, False, False, False], device=dev)

    def segs(p):
        inc = (torch.rand(b, n, generator=g) < p).long(); inc[:, 0] = 0
        return ((4 + inc.cumsum(1)).to(dev) * mask)
    seg = [None, segs(1 / 8), segs(1 / 3), None]                 # mean, two segment levels, one latent per note
    S = [1, int(seg[1].max()) + 1, int(seg[2].max()) + 1, n]
    given = [None] + [(torch.rand(b, s, generator=g) < 0.3).to(dev) for s in S[1:]]
    lmasks = [torch.ones(b, s, dtype=torch.bool, device=dev) for s in S]
    emb = torch.randn(b, n, sum(Ls), device=dev).requires_grad_(True)
    levels = [(seg[i], lmasks[i], S[i], Ls[i], 0.0, given[i]) for i in range(4)]
    out, drop = F_.LatentDropFn.apply(emb, mask, deadpan, levels, inclusive)
    # tensor-op formulation (the product's own path of rounds 1-5, mmd_transformer.py:249-253,275-283)
    level_masks, prior = [], None
    for i in range(4):
        if given[i] is None:
            note = torch.zeros(b, n, dtype=torch.bool, device=dev)
        elif seg[i] is None:
            note = given[i].view(b, n)
        else:
            note = torch.gather(given[i], 1, seg[i])
        if inclusive:
            prior = note if prior is None else (prior | note)
            note = prior
        level_masks.append(note[..., None].expand(b, n, Ls[i]))
    want_drop = torch.cat(level_masks, dim=-1) & mask[..., None] & (~deadpan[:, None, None])
    assert torch.equal(drop, want_drop) and bool(drop.any())
    assert torch.equal(out, emb.detach() * (~want_drop))
    gout = torch.randn_like(out)
    out.backward(gout)
    assert torch.equal(emb.grad, gout * (~want_drop))


def test_latent_drop_draws_at_the_configured_rate_per_latent(dev):
    from scoreperformer_amd import ops
    b, n, S = 64, 240, 40
    seg = (torch.arange(n, device=dev)[None] * S // n).expand(b, n).contiguous()
    lmask = torch.ones(b, S, dtype=torch.bool, device=dev)
    lmask[:, -3:] = False                                        # invalid latents are never dropped
    mask = torch.ones(b, n, dtype=torch.bool, device=dev)
    emb = torch.ones(b, n, 4, device=dev)
    out, drop = ops.latent_drop(emb, mask, None, [(seg, lmask, S, 4, 0.3, None)], True, seed=99)
    per_latent = drop[..., 0].view(b, S, n // S)
    assert bool((per_latent == per_latent[..., :1]).all())       # constant within a segment, all columns alike
    assert bool((drop == drop[..., :1]).all())
    lat_drop = per_latent[..., 0]
    assert not bool(lat_drop[:, -3:].any())
    rate = float(lat_drop[:, :-3].float().mean())
    assert abs(rate - 0.3) < 0.03, rate
    out2, drop2 = ops.latent_drop(emb, mask, None, [(seg, lmask, S, 4, 0.3, None)], True, seed=100)
    assert not torch.equal(drop, drop2)


@pytest.mark.parametrize("hierarchical", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_one_pass_hierarchical_heads_equal_the_level_by_level_path(dev, hierarchical, dtype):
    """HierLatentsFn with ONE pass over the hidden states for all levels (ops.segment_sum_multi / segment_gather_multi) against its
    level-by-level formulation (the path of rounds 1-5, pinned to the reference's fixtures by tests/test_model_gpu.py): same latent
    masks, embeddings / latents / every gradient equal to fp32 summation order (1e-5 of each tensor's scale), ragged batch."""
    from scoreperformer_amd.models.scoreperformer import mmd_transformer as M
    b, n, d = 6, 200, 64
    Ls = [8, 12, 4, 4]
    modes = ("mean", "bar_mean", "beat_mean", "onset_mean")
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(n // 2, n + 1, (b,), generator=g)
    mask = (torch.arange(n)[None] < lens[:, None]).to(dev)

    def segs(p):
        inc = (torch.rand(b, n, generator=g) < p).long(); inc[:, 0] = 0
        return (4 + inc.cumsum(1)).to(dev) * mask
    seg = (None, segs(1 / 16), segs(1 / 4), segs(1 / 2))
    sizes = (1,) + tuple(int(s.max()) + 1 for s in seg[1:])
    hidden0 = torch.randn(b, n, d, generator=g).to(dev).to(dtype)
    d_ins = [d + sum(Ls[:i]) if hierarchical else d for i in range(4)]
    Ws = [torch.randn(L, di, generator=g).to(dev) * 0.1 for L, di in zip(Ls, d_ins)]
    bs = [torch.randn(L, generator=g).to(dev) * 0.1 for L in Ls]
    w_emb = torch.randn(b, n, sum(Ls), generator=g).to(dev)
    w_lat = [torch.randn(b, S, L, generator=g).to(dev) for S, L in zip(sizes, Ls)]
    runs = []
    for one_pass in (False, True):
        M.ONE_PASS_LEVELS = one_pass
        try:
            h = hidden0.clone().requires_grad_(True)
            W = [w.clone().requires_grad_(True) for w in Ws]
            B = [x.clone().requires_grad_(True) for x in bs]
            outs = M.HierLatentsFn.apply(h, mask, hierarchical, modes, seg, sizes, *W, *B)
            emb, lats, lmasks = outs[0], outs[1:5], outs[5:]
            ((emb * w_emb).sum() + sum((l * w).sum() for l, w in zip(lats, w_lat))).backward()
            runs.append((emb.detach(), [l.detach() for l in lats], lmasks, h.grad, [w.grad for w in W], [x.grad for x in B]))
        finally:
            M.ONE_PASS_LEVELS = True

    def close(a, b_):
        return float((a.float() - b_.float()).abs().max()) <= 1e-5 * max(1e-6, float(b_.float().abs().max()))
    (e0, l0, m0, gh0, gw0, gb0), (e1, l1, m1, gh1, gw1, gb1) = runs
    assert all(torch.equal(a, b_) for a, b_ in zip(m0, m1)) and any(bool((~m).any()) for m in m0[1:])
    assert close(e1, e0) and all(close(a, b_) for a, b_ in zip(l1, l0))
    assert gh1.dtype == gh0.dtype and close(gh1, gh0) and float(gh1[~mask].abs().max()) == 0.0
    assert all(close(a, b_) for a, b_ in zip(gw1, gw0)) and all(close(a, b_) for a, b_ in zip(gb1, gb0))
