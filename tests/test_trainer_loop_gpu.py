"""GPU: the model driven EXACTLY the way the reference's trainer drives it -- no arena, no fused optimizer, no edit to the loop:

    model.to(device)                                             trainer.py:129
    torch.optim.AdamW(model.parameters(), ...)                   experiments/optimizers.py (the wrapped optimizer)
    for micro-batch in accumulation window:                      trainer.py:449-456, grad_accum_steps > 1
        scaler.scale(loss / grad_accum_steps).backward()         optimizers.py:151-158
    scaler.unscale_(opt); clip_grad_norm_(params, max_norm)      optimizers.py:159-164
    scaler.step(opt); scaler.update(); opt.zero_grad()           optimizers.py:165-169

against the fp32 CPU oracle running the same accumulation window (two backward passes into one gradient, torch-equivalent clip +
AdamW), and against the fast binding (ParamArena + FusedAdamW) driven through the same window, which pins gradient ACCUMULATION across
two backward calls on the arena path as well.  Dropout 0, the N(0, I) samples of the MMD term injected on both sides.

Both loops also run in the reference's DEFAULT precision mode (`recipes/default.yaml:89` `mixed_precision: true`): the forward inside
`torch.cuda.amp.autocast(enabled=True)` (`trainer.py:449`) and an enabled `GradScaler` (`optimizers.py:146`: loss x 65 536, `unscale_`,
inf-skip).  On this path autocast decides nothing -- every public forward fences it off (`utils/amp.py`), bf16 operands regardless --
so the bounds are those of the unscaled loop, plus: the scaler must not skip a step (its scale stays at 65 536: no inf / nan reached
a gradient) and the unscaled gradients equal the ones of the plain loop to bf16 rounding of the scaled values."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
LR, WD, CLIP, ACCUM = 5e-4, 1e-6, 2.0, 2


def _uniq_names(model):
    seen, names = set(), []
    for n, p in model.named_parameters():
        if id(p) not in seen:
            seen.add(id(p)); names.append(n)
    return names


def _oracle_leaves(sd0):
    """The oracle's flat state dict with ONE leaf per tied tensor (gradients accumulate like in the module tree)."""
    from oracle.weights import canonical
    leaves, sdg = {}, {}
    for k, v in sd0.items():
        leaf = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v.clone()
        sdg[k] = leaves.setdefault(canonical(k), leaf)
    return sdg


def _oracle_window(sdg, cfg, batches, zs, names, state, step):
    """One accumulation window on the CPU: len(batches) backward passes of loss / ACCUM into the same leaves, then clip + AdamW."""
    from oracle import ref_cpu
    params = [sdg[n] for n in names]
    for p in params:
        p.grad = None
    losses = []
    for batch, z in zip(batches, zs):
        out = ref_cpu.score_performer_forward(sdg, cfg, batch, z, training=True)
        (out["loss"] / ACCUM).backward()
        losses.append(float(out["loss"].detach()))
    grads = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    with torch.no_grad():
        ref_cpu.clip_adamw_step(params, grads, state["m"], state["v"], step, lr=LR, weight_decay=WD, max_norm=CLIP)
    return losses, grads


def _reference_window(model, opt, scaler, batches, zs, dev, amp=False):
    """trainer.py:449-456 + optimizers.py:151-169, verbatim in shape: nothing here knows about the HIP path."""
    from torch.cuda import amp as cuda_amp                       # trainer.py:19 / optimizers.py:7
    losses = []
    for batch, z in zip(batches, zs):
        model.perf_encoder._z_override = [t.to(dev) for t in z]
        with cuda_amp.autocast(enabled=amp):                     # trainer.py:449 (fp16 autocast, the CUDA default)
            assert torch.is_autocast_enabled("cuda") == amp
            out = model(**batch)
        assert out.loss.dtype == torch.float32
        scaler.scale(out.loss / ACCUM).backward()
        losses.append(float(out.loss.detach()))
    scaler.unscale_(opt)
    params = [p for g in opt.param_groups for p in g["params"]]
    grads = {id(p): (p.grad.detach().clone() if p.grad is not None else None) for p in params}
    torch.nn.utils.clip_grad_norm_(params, CLIP)
    scaler.step(opt)
    scaler.update()
    opt.zero_grad()
    return losses, grads


def _setup(preset, dev, seq, n_windows, seed, own_init=False):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config(preset, dropout=0.0)
    torch.manual_seed(1234)
    model = ScorePerformer.init(model_config(preset, dropout=0.0))
    if not own_init:
        model.load_state_dict(filled_state_dict(model, seed=seed))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batches = [[synthetic_batch(2 if preset == "tiny" else 1, seq, seed=seed + 10 * w + a, ragged=True) for a in range(ACCUM)] for w in range(n_windows)]
    zs = [[[torch.randn(256, d, generator=torch.Generator().manual_seed(1000 * w + 100 * a + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
           for a in range(ACCUM)] for w in range(n_windows)]
    gb = [[{k: v.to(dev) for k, v in b.items()} for b in win] for win in batches]
    return cfg, model, sd0, batches, gb, zs


@pytest.mark.parametrize("amp", [False, True], ids=["fp32-loop", "mixed_precision"])
def test_reference_trainer_loop_without_arena_matches_the_oracle_and_the_arena_binding(dev, amp):
    """Tiny model, three accumulation windows of two micro-batches.  (1) The zero-edit binding (model.to + torch.optim.AdamW +
    GradScaler(enabled=False)) follows the oracle: every micro-batch loss within 5e-3 relative (the bound of
    test_train_trajectory_matches_cpu_oracle), the ACCUMULATED gradient of the first window per tensor within 6 % + 2e-3 in norm.
    (2) The fast binding (ParamArena + FusedAdamW, two backward calls, one step) lands on the same parameters: the update of every
    window as one vector within 2 % relative L2 of the zero-edit binding's."""
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    n_windows = 3
    cfg, model, sd0, batches, gb, zs = _setup("tiny", dev, 64, n_windows, seed=31)
    names = _uniq_names(model)

    # ---- (1) the reference's loop, untouched
    model.to(dev)
    assert all(p.is_cuda and getattr(p, "_spn_main_grad", None) is None for p in model.parameters())     # no arena behind it
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=WD)
    scaler = torch.cuda.amp.GradScaler(enabled=amp)               # optimizers.py:146
    sdg = _oracle_leaves(sd0)
    state = {"m": [torch.zeros_like(sdg[n]) for n in names], "v": [torch.zeros_like(sdg[n]) for n in names]}
    named = dict(model.named_parameters())
    traj = []
    for w in range(n_windows):
        before = torch.cat([named[n].detach().float().reshape(-1) for n in names])
        got, ggrads = _reference_window(model, opt, scaler, gb[w], zs[w], dev, amp)
        want, ograds = _oracle_window(sdg, cfg, batches[w], zs[w], names, state, w + 1)
        if amp:
            assert scaler.get_scale() == 65536.0, (w, scaler.get_scale())                                # no step was skipped
        for a, b in zip(got, want):
            assert abs(a - b) <= 5e-3 * abs(b), (w, got, want)
        if w == 0:
            bad = []
            for n, og in zip(names, ograds):
                g = ggrads[id(named[n])]
                if g is None:
                    assert float(og.abs().max()) == 0.0, n       # the product skips exactly the tensors the oracle has no gradient for
                    continue
                gn, on = float(g.double().norm()), float(og.double().norm())
                if abs(gn - on) > 0.06 * on + 2e-3:
                    bad.append((n, gn, on))
            assert not bad, bad[:10]
        traj.append(torch.cat([named[n].detach().float().reshape(-1) for n in names]) - before)
        assert all(p.grad is None for p in model.parameters())                                           # opt.zero_grad() (set_to_none)
    assert all(torch.isfinite(t).all() for t in traj)
    assert all(float(t.abs().max()) > 0 for t in traj)                                                   # every window stepped

    # ---- (2) the fast binding through the same windows
    model2 = ScorePerformer.init(model_config("tiny", dropout=0.0))
    model2.load_state_dict(sd0)
    arena = ParamArena(model2, dev)
    model2.train()
    fused = FusedAdamW(arena, lr=LR, weight_decay=WD, grad_clip=CLIP)
    named2 = dict(model2.named_parameters())
    for w in range(n_windows):
        before = torch.cat([named2[n].detach().float().reshape(-1) for n in names])
        for batch, z in zip(gb[w], zs[w]):
            model2.perf_encoder._z_override = [t.to(dev) for t in z]
            (model2(**batch).loss / ACCUM).backward()            # accumulates into the arena's gradient buffer
        fused.step()                                             # clip + AdamW + zero_grad
        delta = torch.cat([named2[n].detach().float().reshape(-1) for n in names]) - before
        rel = float((delta - traj[w]).norm() / traj[w].norm())
        assert rel <= 2e-2, (w, rel)
        assert float(arena.grads.abs().max()) == 0.0


def test_two_accumulating_backward_calls_equal_one_backward_of_the_mean_loss(dev):
    """Gradient accumulation on both bindings, without the oracle: backward(loss_a / 2) then backward(loss_b / 2) leaves the same gradient
    as ONE backward of (loss_a + loss_b) / 2 (relative L2 <= 2e-3: the same kernels on the same values, other summation orders and one
    bf16 rounding of the 1/2 factor's position)."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    cfg, model, sd0, batches, gb, zs = _setup("tiny", dev, 64, 1, seed=5)
    names = _uniq_names(model)

    def grads_of(m, use_arena, split):
        named = dict(m.named_parameters())
        if use_arena:
            arena = m._test_arena
            arena.zero_grad()
        else:
            m.zero_grad(set_to_none=True)
        outs = []
        for batch, z in zip(gb[0], zs[0]):
            m.perf_encoder._z_override = [t.to(dev) for t in z]
            out = m(**batch)
            if split:
                (out.loss / ACCUM).backward()
            else:
                outs.append(out.loss)
        if not split:
            (sum(outs) / ACCUM).backward()
        return torch.cat([(named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])).detach().float().reshape(-1).clone()
                          for n in names])

    for use_arena in (False, True):
        m = ScorePerformer.init(model_config("tiny", dropout=0.0))
        m.load_state_dict(sd0)
        if use_arena:
            m._test_arena = ParamArena(m, dev)
        else:
            m.to(dev)
        m.train()
        a, b = grads_of(m, use_arena, True), grads_of(m, use_arena, False)
        rel = float((a - b).norm() / b.norm())
        assert rel <= 2e-3, (use_arena, rel)


@pytest.mark.parametrize("amp", [False, True], ids=["fp32-loop", "mixed_precision"])
def test_reference_trainer_loop_at_c2_scale(dev, amp):
    """The C2 model (d = 512, 6/6/6 layers, 71.9 M parameters at its own initialisation), one accumulation window of two micro-batches of
    1 x 1024 notes through the zero-edit binding: each micro-batch loss within 1e-3 of the oracle (north_star), the accumulated gradient
    as one vector within 5 % relative L2 and 1 % in norm (the bounds of test_c3_step_matches_oracle), and the AdamW update moves the
    parameters in the oracle's direction (cosine of the two updates >= 0.9; Adam's first step is sign-like, so bf16 noise on near-zero
    gradient entries flips individual signs)."""
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    cfg, model, sd0, batches, gb, zs = _setup("c2", dev, 1024, 1, seed=41, own_init=True)
    names = _uniq_names(model)
    model.to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=LR, weight_decay=WD)
    scaler = torch.cuda.amp.GradScaler(enabled=amp)
    named = dict(model.named_parameters())
    before = [named[n].detach().float().cpu().clone() for n in names]
    got, ggrads = _reference_window(model, opt, scaler, gb[0], zs[0], dev, amp)
    if amp:
        assert scaler.get_scale() == 65536.0
    sdg = _oracle_leaves(sd0)
    state = {"m": [torch.zeros_like(sdg[n]) for n in names], "v": [torch.zeros_like(sdg[n]) for n in names]}
    o_before = [sdg[n].detach().clone() for n in names]
    want, ograds = _oracle_window(sdg, cfg, batches[0], zs[0], names, state, 1)
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-3, (got, want)
    err2 = ref2 = got2 = 0.0
    for n, og in zip(names, ograds):
        g = ggrads[id(named[n])]
        g = torch.zeros_like(og) if g is None else g.float().cpu()
        err2 += float((g - og).double().pow(2).sum()); ref2 += float(og.double().pow(2).sum()); got2 += float(g.double().pow(2).sum())
    assert err2 ** 0.5 <= 0.05 * ref2 ** 0.5, (err2 ** 0.5, ref2 ** 0.5)
    assert abs(got2 ** 0.5 - ref2 ** 0.5) <= 1e-2 * ref2 ** 0.5
    du = torch.cat([(named[n].detach().float().cpu() - b0).reshape(-1) for n, b0 in zip(names, before)])
    do = torch.cat([(sdg[n].detach() - b0).reshape(-1) for n, b0 in zip(names, o_before)])
    cos = float(torch.dot(du, do) / (du.norm() * do.norm()))
    assert cos >= 0.9, cos


def test_dropout_of_a_width_off_the_8_element_grid(dev):
    """The stand-alone dropout takes any width, like nn.Dropout: widths off the kernel's grid run through a padded flat
    copy -- kept entries scaled by 1 / (1 - p), rate p, the same seed reproduces the mask (that is the backward), eval is the identity."""
    from scoreperformer_amd import functional as F_, ops
    x = torch.randn(37, 13, 21, device=dev)
    y = ops.dropout(x, 0.25, seed=1234)
    assert y.shape == x.shape
    kept = y != 0
    assert torch.allclose(y[kept], x[kept] / 0.75, rtol=1e-6, atol=0)
    assert abs(float((~kept).float().mean()) - 0.25) < 0.03
    assert torch.equal(ops.dropout(x, 0.25, seed=1234), y) and not torch.equal(ops.dropout(x, 0.25, seed=1235), y)
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(3)
    out = F_.dropout(xr, 0.25, training=True)
    out.sum().backward()
    assert torch.equal(xr.grad != 0, out != 0)
    assert F_.dropout(xr, 0.25, training=False) is xr


def test_dropout_mask_does_not_depend_on_the_strides_of_its_argument(dev):
    """the forward (on x) and the backward (on dy) pick the mask's definition independently, so it may depend on the WIDTH only.
    A narrowed view whose row stride is off the kernel's 8-element grid, the same values packed, and a column slice starting off the
    16-byte grid all drop the same entries; through autograd, a strided x with a contiguous dy gets the forward's mask."""
    from scoreperformer_amd import functional as F_, ops
    base = torch.randn(64, 44, device=dev)
    view = base[:, :40]                                   # width 40 (on the grid), row stride 44 (off it)
    assert view.stride(0) % 8 and view.shape[1] % 8 == 0
    a, b = ops.dropout(view, 0.3, seed=77), ops.dropout(view.contiguous(), 0.3, seed=77)
    assert torch.equal(a, b)
    off = torch.randn(64, 52, device=dev)[:, 4:44]        # stride 52, first column 4: rows start off the 16-byte grid
    assert torch.equal(ops.dropout(off, 0.3, seed=78), ops.dropout(off.contiguous(), 0.3, seed=78))
    xr = base.clone().requires_grad_(True)
    torch.manual_seed(9)
    y = F_.dropout(xr[:, :40], 0.3, training=True)        # forward on the strided view
    y.backward(torch.ones(64, 40, device=dev))            # backward on a packed gradient
    assert torch.equal(xr.grad[:, :40] != 0, y != 0) and float(xr.grad[:, 40:].abs().max()) == 0.0
