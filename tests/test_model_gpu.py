"""GPU parity of the full HIP-backed model against the CPU oracle on the golden configurations, plus the committed
reference outputs (tests/golden).  Tolerances: bf16 operands / fp32 accumulation vs an fp32 CPU path.
north_star: loss within 1e-3 ... we assert 2e-2 relative on tiny random-weight models here (few tokens, no averaging) and
check the 1e-3 figure at benchmark scale in bench.py's parity leg."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SMALL_VOCAB = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10,
               "PositionShift": 21, "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}
VARIANTS = {
    "tiny_mixlm": dict(preset="tiny", num_tokens=SMALL_VOCAB),
    "tiny_xattn_mha": dict(preset="tiny", context_emb_mode="attention", style_emb_mode="cat", one_kv_head=False,
                           alibi_learned=False, num_tokens=SMALL_VOCAB),
    "tiny_full_vocab": dict(preset="tiny"),
}


def build(name, dev, seed=0):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    model = ScorePerformer.init(model_config(**VARIANTS[name]))
    sd = filled_state_dict(model, seed=seed)
    model.load_state_dict(sd, strict=True)
    arena = ParamArena(model, dev)
    return model, arena, sd


@pytest.mark.parametrize("name", list(VARIANTS))
def test_forward_backward_matches_reference_golden(dev, name):
    fix = dict(np.load(os.path.join(GOLD, f"{name}.npz"), allow_pickle=False))
    model, arena, _ = build(name, dev)
    model.train()
    batch = {k[3:]: torch.from_numpy(v).to(dev) for k, v in fix.items() if k.startswith("in/")}
    model.perf_encoder._z_override = [torch.from_numpy(fix[f"z/{i}"]).to(dev) for i in range(4)]
    out = model(**batch)
    ref_losses = {k[7:]: float(v) for k, v in fix.items() if k.startswith("losses/")}
    assert set(out.losses) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(float(out.losses[k]) - v) <= 2e-2 * max(1.0, abs(v)), (k, float(out.losses[k]), v)
    assert abs(float(out.loss) - float(fix["out/loss"])) <= 2e-2 * float(fix["out/loss"])
    for k, v in fix.items():
        if k.startswith("logits/"):
            got = out.perf_decoder.logits[k[7:]].detach().float().cpu().numpy()
            assert np.abs(got - v).max() <= 0.03 * np.abs(v).max(), k
    hs = out.perf_decoder.hidden_state.detach().float().cpu().numpy()
    assert np.abs(hs - fix["out/hidden_state"]).max() <= 0.03 * np.abs(fix["out/hidden_state"]).max()
    pe = out.perf_encoder.embeddings.detach().float().cpu().numpy()
    assert np.abs(pe - fix["out/perf_embeddings"]).max() <= 0.03 * np.abs(fix["out/perf_embeddings"]).max() + 1e-3

    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    bad = []
    for k, v in fix.items():
        if k.startswith("gradnorm/"):
            g = named[k[9:]].grad
            got = float(g.double().norm())
            if abs(got - float(v)) > 0.06 * float(v) + 2e-3:
                bad.append((k[9:], got, float(v)))
    assert not bad, bad[:10]
    for k, v in fix.items():
        if k.startswith("grad/"):
            g = named[k[5:]].grad.float().cpu().numpy()
            # the ALiBi log-slope gradient is a signed sum of distance-weighted score gradients over every query/key pair: what
            # survives the cancellation is ~5e-4 here and the bf16 noise on it measures 1.4e-4 .. 2.2e-4 (tools history: a 1-ulp
            # fp32 change in SiLU moves it by 2e-5), so its absolute floor is 3e-4; every other tensor sits below 0.25 of its bound
            floor = 3e-4 if k.endswith("learned_logslopes") else 1e-4
            assert np.abs(g - v).max() <= 0.06 * np.abs(v).max() + floor, k


def test_optimizer_step_matches_torch_adamw(dev):
    from scoreperformer_amd.arena import FusedAdamW
    fix = dict(np.load(os.path.join(GOLD, "tiny_mixlm.npz"), allow_pickle=False))
    model, arena, sd = build("tiny_mixlm", dev)
    g = torch.Generator().manual_seed(0)
    # synthetic gradients, same on both sides
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in arena.param_list]
    for p, rp in zip(arena.param_list, ref_params):
        gr = torch.randn(p.shape, generator=g) * 0.3
        p.grad.copy_(gr.to(dev))
        rp.grad = gr.clone()
    total = torch.nn.utils.clip_grad_norm_(ref_params, 2.0)
    opt_ref = torch.optim.AdamW(ref_params, lr=2e-4, weight_decay=1e-6)
    opt_ref.step()
    opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-6, grad_clip=2.0)
    norm = opt.step()
    assert abs(float(norm) - float(total)) <= 1e-4 * float(total)
    for p, rp in zip(arena.param_list, ref_params):
        assert (p.detach().cpu() - rp.detach()).abs().max().item() <= 2e-7 + 1e-6 * rp.abs().max().item()
    assert arena.grads.abs().max().item() == 0
    # bf16 compute copy refreshed
    p0 = arena.param_list[5]
    assert (p0._spn_shadow.float() - p0.detach()).abs().max().item() <= 2 ** -8 * p0.abs().max().item()


def test_optimizer_keeps_a_step_count_per_parameter_like_torch_adamw(dev):
    """A parameter that is frozen for some steps (Model.freeze, models/base.py:95-102) or simply gets no gradient falls behind: torch.optim.AdamW
    bias-corrects it with ITS OWN step count and stores that count per parameter.  Five steps with a changing set of gradient-less
    parameters follow torch's trajectory, the state dict carries torch's per-parameter steps, and a torch optimizer state with differing
    steps loads back."""
    from scoreperformer_amd.arena import FusedAdamW, ParamArena
    model, arena, sd = build("tiny_mixlm", dev)
    g = torch.Generator().manual_seed(1)
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in arena.param_list]
    opt_ref = torch.optim.AdamW(ref_params, lr=1e-2, weight_decay=1e-2)
    opt = FusedAdamW(arena, lr=1e-2, weight_decay=1e-2, grad_clip=None)
    n = len(ref_params)
    for step in range(5):
        off = {i for i in range(n) if (i + step) % 3 == 0 and step in (1, 2, 4)} | ({0, 1, 2} if step < 3 else set())
        for i, (p, rp) in enumerate(zip(arena.param_list, ref_params)):
            if i in off:
                rp.grad = None
                p._spn_touched = False
                continue
            gr = torch.randn(p.shape, generator=g) * 0.3
            p.grad.copy_(gr.to(dev))
            p._spn_touched = True
            rp.grad = gr.clone()
        opt_ref.step()
        opt.step()
    assert len(set(arena.steps)) > 2 and arena.steps[0] == 2
    for i, (p, rp) in enumerate(zip(arena.param_list, ref_params)):
        assert (p.detach().cpu() - rp.detach()).abs().max().item() <= 1e-6 + 2e-5 * rp.abs().max().item(), (i, arena.steps[i])
    ours, theirs = opt.state_dict(), opt_ref.state_dict()
    assert sorted(ours["state"]) == sorted(theirs["state"])
    for i, st in theirs["state"].items():
        assert int(ours["state"][i]["step"]) == int(st["step"]) == arena.steps[i]
        assert (ours["state"][i]["exp_avg"].cpu() - st["exp_avg"]).abs().max().item() <= 1e-6
    # a torch state with differing steps loads (a reference optimizer checkpoint taken after a freeze / unfreeze)
    model2, arena2, _ = build("tiny_mixlm", dev)
    opt2 = FusedAdamW(arena2, lr=1.0)
    opt2.load_state_dict(theirs)
    assert arena2.steps == arena.steps and opt2.lr == 1e-2


def test_greedy_render_matches_reference_tokens(dev):
    """Cached greedy `unmask_tokens` on the GPU reproduces the reference's tokens bit-exactly (north_star: bit-exact token
    argmax at greedy decode); fixture: 39 notes x 4 predicted dims from the reference's own cached decode."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy.npz"), allow_pickle=False))
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB))
    model.load_state_dict(filled_state_dict(model, seed=3))
    ParamArena(model, dev)
    model.eval()
    batch = {k[3:]: torch.from_numpy(v).to(dev) for k, v in fix.items() if k.startswith("in/") and k != "in/tokens"}
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"],
                                     score_mask=batch["score_mask"], bars=batch["bars"], beats=batch["beats"],
                                     onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"], compute_loss=False)
    se = enc.score_embeddings.float().cpu().numpy()
    assert np.abs(se - fix["out/score_embeddings"]).max() <= 0.03 * np.abs(fix["out/score_embeddings"]).max()
    tokens = torch.from_numpy(fix["in/tokens"]).to(dev)
    out, caches = model.perf_decoder.unmask_tokens(
        tokens, batch["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
        filter_logits_fn=top_k, filter_kwargs={"k": 1}, return_caches=True, disable_tqdm=True)
    want = fix["out/tokens"]
    got = out.cpu().numpy()
    mism = np.argwhere(got != want)
    # bf16 operands cannot reproduce an fp32 arg-max where the reference's own top-2 margin is below bf16 resolution:
    # every differing token must be such a near-tie (margin from the CPU oracle, teacher-forced on the reference tokens)
    if len(mism):
        from oracle import ref_cpu
        cfg = model_config(preset="tiny", num_tokens=SMALL_VOCAB)
        sd = filled_state_dict(ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB)), seed=3)
        ref_tok = torch.from_numpy(want)
        masked = torch.from_numpy(fix["in/masked_perf"])
        ctx = torch.from_numpy(fix["out/score_embeddings"])[:, 1:]
        sty = torch.from_numpy(fix["out/perf_embeddings"])[:, 1:]
        _, logits = ref_cpu.tuple_transformer(sd, "perf_decoder.model.", cfg["perf_decoder"], [ref_tok[:, :-1], masked[:, 1:]],
                                              causal=True, mask=torch.ones(1, ref_tok.shape[1] - 1, dtype=torch.bool),
                                              context=ctx, style=sty, with_logits=True)
        keys = list(logits.keys())
        first = mism[np.lexsort((mism[:, 2], mism[:, 1]))][0]   # earliest differing position (later ones may be induced)
        _, pos, dim = first
        lg = logits[keys[dim]][0, pos - 1].clone()
        lg[:2] = -float("inf")
        top2 = torch.topk(lg, 2)
        margin = float(top2.values[0] - top2.values[1])
        assert int(top2.indices[0]) == int(want[0, pos, dim])
        assert got[0, pos, dim] == int(top2.indices[1]) and margin < 0.05 * float(lg[lg > -1e30].abs().max()), (first, margin)
    assert len(mism) <= 0.03 * (want != fix["in/tokens"]).sum(), f"{len(mism)} tokens differ"
    assert tuple(caches.token_emb.shape) == tuple(fix["cache/token_emb_shape"])
    assert len(caches.transformer.hiddens) == int(fix["cache/n_hiddens"])


def test_decode_engine_matches_reference_and_module_path(dev):
    """The hipGraph-replayed fp32 decode engine: tokens equal to the reference's cached decode (fixture) bit for bit, and its
    caches equal to the module path's caches."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy.npz"), allow_pickle=False))
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB))
    model.load_state_dict(filled_state_dict(model, seed=3))
    ParamArena(model, dev)
    model.eval()
    tokens = torch.from_numpy(fix["in/tokens"]).to(dev)
    masked = torch.from_numpy(fix["in/masked_perf"]).to(dev)
    # feed the REFERENCE encoder outputs so that only the decode path is under test
    ctx = torch.from_numpy(fix["out/score_embeddings"]).to(dev)
    sty = torch.from_numpy(fix["out/perf_embeddings"]).to(dev)
    dec = model.perf_decoder
    out_e, caches_e = dec.unmask_tokens(tokens, masked, context=ctx, style_embeddings=sty, filter_logits_fn=top_k,
                                        filter_kwargs={"k": 1}, return_caches=True, disable_tqdm=True)
    assert int((out_e.cpu().numpy() != fix["out/tokens"]).sum()) == 0          # bit-exact greedy tokens
    assert tuple(caches_e.token_emb.shape) == tuple(fix["cache/token_emb_shape"])
    assert len(caches_e.transformer.hiddens) == int(fix["cache/n_hiddens"])
    assert tuple(caches_e.transformer.attention[0].keys.shape) == tuple(fix["cache/keys0_shape"])
    dec.use_decode_engine = False
    out_m, caches_m = dec.unmask_tokens(tokens, masked, context=ctx, style_embeddings=sty, filter_logits_fn=top_k,
                                        filter_kwargs={"k": 1}, return_caches=True, disable_tqdm=True)
    same = (out_m == out_e).float().mean().item()
    assert same > 0.97
    if same == 1.0:
        for a, b_ in zip(caches_e.transformer.hiddens, caches_m.transformer.hiddens):
            assert (a - b_.float()).abs().max().item() <= 0.05 * b_.float().abs().max().item()


def test_load_state_dict_refreshes_bf16_copies(dev):
    """Weights written behind the optimizer's back must reach the bf16 compute copies, including the fused q|k|v views."""
    model, arena, sd = build("tiny_mixlm", dev)
    att = model.perf_decoder.model.transformer.layers[0][1]
    assert att._w_qkv is not None
    sd2 = {k: (v * 1.5 if k.endswith("to_q.weight") else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd2)
    from scoreperformer_amd import functional as F_
    wq = F_.bf16_weight(att._w_qkv)[:att.to_q.weight.shape[0]].float()
    assert (wq - att.to_q.weight.detach()).abs().max().item() <= 2 ** -7 * att.to_q.weight.abs().max().item()
    with torch.no_grad():
        att.to_k.weight.mul_(2.0)            # in-place edit of one constituent
    wk = F_.bf16_weight(att._w_qkv)[att.to_q.weight.shape[0]:att.to_q.weight.shape[0] + att.to_k.weight.shape[0]].float()
    assert (wk - att.to_k.weight.detach()).abs().max().item() <= 2 ** -7 * att.to_k.weight.abs().max().item()


def test_optimizer_state_dict_round_trips_through_torch_adamw_format(dev):
    """FusedAdamW.state_dict() is torch.optim.AdamW's format (reference checkpoints' "optimizer" entry): a torch AdamW loaded
    from it continues exactly like the fused optimizer, and a fresh arena resumed from it does too."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config("tiny", dropout=0.0)
    batch = synthetic_batch(2, 48, seed=2, ragged=True, device=dev)

    def make():
        m = ScorePerformer.init(model_config("tiny", dropout=0.0))
        m.load_state_dict(filled_state_dict(m, seed=9))
        a = ParamArena(m, dev)
        m.train()
        return m, a, FusedAdamW(a, lr=1e-3, weight_decay=1e-2, grad_clip=None)

    def run(m, a, o, n):
        for _ in range(n):
            torch.manual_seed(5)
            a.zero_grad()
            m(**batch).loss.backward()
            o.step()

    m1, a1, o1 = make()
    run(m1, a1, o1, 2)
    sd_model = {k: v.clone() for k, v in m1.state_dict().items()}
    sd_opt = o1.state_dict()
    # torch's own AdamW accepts the dict
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in a1.param_list]
    topt = torch.optim.AdamW(ref_params, lr=1.0)
    topt.load_state_dict(sd_opt)
    assert topt.param_groups[0]["lr"] == 1e-3 and int(topt.state[ref_params[0]]["step"]) == 2
    # resume in a fresh arena and continue: same parameters as the uninterrupted run (up to summation-order noise)
    m2, a2, o2 = make()
    m2.load_state_dict(sd_model)
    o2.load_state_dict(sd_opt)
    run(m1, a1, o1, 1)
    run(m2, a2, o2, 1)
    assert a2.step_count == 3
    assert (a1.params - a2.params).abs().max() <= 2e-5


def test_train_trajectory_matches_cpu_oracle(dev):
    """Six full train steps (forward, backward, clip, AdamW) on the GPU follow the fp32 CPU oracle's loss trajectory:
    the north-star statement 'loss within 1e-3 of the CPU reference' held over consecutive updates, not just at step 0."""
    from oracle import ref_cpu
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config("tiny", dropout=0.0)
    model = ScorePerformer.init(model_config("tiny", dropout=0.0))
    sd0 = filled_state_dict(model, seed=21)
    model.load_state_dict(sd0)
    arena = ParamArena(model, dev)
    model.train()
    lr, wd, clip, steps = 5e-4, 1e-6, 2.0, 6
    opt = FusedAdamW(arena, lr=lr, weight_decay=wd, grad_clip=clip)
    batch = synthetic_batch(2, 64, seed=13, ragged=True)
    gbatch = {k: v.to(dev) for k, v in batch.items()}
    zs = [[torch.randn(256, d, generator=torch.Generator().manual_seed(100 * s + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
          for s in range(steps)]
    # CPU oracle: same weights, same batch, same N(0, I) samples per step, torch-equivalent clip + AdamW
    names = [n for n, _ in model.named_parameters()]
    seen, uniq = set(), []
    for n, p in model.named_parameters():
        if id(p) not in seen:
            seen.add(id(p)); uniq.append(n)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v.clone()) for k, v in sd0.items()}
    # tied parameters share storage in the model's state_dict; make the oracle's dict share tensors the same way
    ptr_of = {}
    for k, v in model.state_dict().items():
        ptr_of.setdefault(v.data_ptr(), k)
    for k, v in model.state_dict().items():
        first = ptr_of[v.data_ptr()]
        if first != k:
            sd[k] = sd[first]
    params = [sd[n] for n in uniq]
    m = [torch.zeros_like(p) for p in params]
    v2 = [torch.zeros_like(p) for p in params]
    cpu_losses, gpu_losses = [], []
    for s in range(steps):
        for p in params:
            p.grad = None
        out = ref_cpu.score_performer_forward(sd, cfg, batch, zs[s], training=True)
        out["loss"].backward()
        cpu_losses.append(float(out["loss"].detach()))
        with torch.no_grad():
            ref_cpu.clip_adamw_step(params, [p.grad if p.grad is not None else torch.zeros_like(p) for p in params], m, v2, s + 1,
                                    lr=lr, weight_decay=wd, max_norm=clip)
        model.perf_encoder._z_override = [t.to(dev) for t in zs[s]]
        arena.zero_grad()
        gout = model(**gbatch)
        gout.loss.backward()
        gpu_losses.append(float(gout.loss.detach()))
        opt.step()
    # bf16 GEMM operands give ~1e-3 relative on this random-weight tiny model at step 0; Adam's first updates are sign-like
    # (|update| ~ lr whatever the gradient's size), so that noise is carried, not amplified: 5e-3 relative bounds every step
    for s, (a, b) in enumerate(zip(gpu_losses, cpu_losses)):
        assert abs(a - b) <= 5e-3 * abs(b), (s, gpu_losses, cpu_losses)
    assert cpu_losses[-1] < cpu_losses[0] and gpu_losses[-1] < gpu_losses[0]   # and both actually learn


def test_decode_with_cross_attending_decoder_matches_the_oracle(dev):
    """context_emb_mode='attention' (decoder blocks ('a','c','f')): the hipGraph decode engine AND the module path with caches give the
    tokens of the CPU oracle's greedy loop on the fixture's inputs (padded score -> context mask).  The oracle is pinned to the
    reference's own cached decode in this mode (tests/test_oracle_golden.py), whose tokens it reproduces bit for bit once the
    reference's stale-hidden-row defect is switched on; here the intended row is used on both sides."""
    from oracle import ref_cpu
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy_xattn.npz"), allow_pickle=False))
    kw = dict(preset="tiny", context_emb_mode="attention", num_tokens=SMALL_VOCAB)
    cfg = model_config(**kw)
    model = ScorePerformer.init(model_config(**kw))
    sd = filled_state_dict(model, seed=4)
    model.load_state_dict(sd)
    ParamArena(model, dev)
    model.eval()
    tokens, masked = torch.from_numpy(fix["in/tokens"]), torch.from_numpy(fix["in/masked_perf"])
    ctx, sty = torch.from_numpy(fix["out/score_embeddings"]), torch.from_numpy(fix["out/perf_embeddings"])
    cmask = torch.from_numpy(fix["in/score_mask"])
    want = ref_cpu.greedy_unmask(sd, cfg, tokens, masked, ctx, sty, context_mask=cmask).numpy()
    assert (want[:, :3] == fix["out/tokens"][:, :3]).all()          # the notes the reference decodes before its defect bites
    dec = model.perf_decoder
    args = dict(context=ctx.to(dev), context_mask=cmask.to(dev), style_embeddings=sty.to(dev), filter_logits_fn=top_k, filter_kwargs={"k": 1},
                return_caches=True, disable_tqdm=True)
    out_e, caches_e = dec.unmask_tokens(tokens.to(dev), masked.to(dev), **args)
    assert int((out_e.cpu().numpy() != want).sum()) == 0              # fp32 engine: bit-exact greedy tokens
    assert len(caches_e.transformer.attention) == 4 and len(caches_e.transformer.hiddens) == 3     # a, c, a, c + final
    assert caches_e.transformer.attention[1].keys.shape[-2] == ctx.shape[1]
    dec.use_decode_engine = False
    out_m, _ = dec.unmask_tokens(tokens.to(dev), masked.to(dev), **args)
    assert (out_m.cpu().numpy() == want).mean() > 0.97                # bf16 GEMMs: only near-ties of the top-2 logits may flip


def _xattn_model(dev):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    kw = dict(preset="tiny", context_emb_mode="attention", num_tokens=SMALL_VOCAB)
    model = ScorePerformer.init(model_config(**kw))
    sd = filled_state_dict(model, seed=4)
    model.load_state_dict(sd)
    ParamArena(model, dev)
    model.eval()
    return model, sd, model_config(**kw)


def test_reference_compat_reproduces_the_references_cross_attention_decode_token_for_token(dev):
    """`reference_compat=True`: the decode engine of a cross-attending decoder reads the final hidden of the row the REFERENCE reads
    (wrappers.py:364 after the per-prefix rows of modules/transformer/transformer.py:201 have piled up in its caches) and so reproduces the
    reference's own cached decode of tests/golden/tiny_greedy_xattn.npz -- all 39 notes x 4 predicted dims, not only the first three notes
    that precede the defect.  (Default False: the intended row, test_decode_with_cross_attending_decoder_matches_the_oracle.)"""
    from scoreperformer_amd.modules.sampling import top_k
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy_xattn.npz"), allow_pickle=False))
    model, sd, cfg = _xattn_model(dev)
    dec = model.perf_decoder
    tokens, masked = torch.from_numpy(fix["in/tokens"]).to(dev), torch.from_numpy(fix["in/masked_perf"]).to(dev)
    args = dict(context=torch.from_numpy(fix["out/score_embeddings"]).to(dev), context_mask=torch.from_numpy(fix["in/score_mask"]).to(dev),
                style_embeddings=torch.from_numpy(fix["out/perf_embeddings"]).to(dev), filter_logits_fn=top_k, filter_kwargs={"k": 1},
                disable_tqdm=True)
    dec.reference_compat = True
    try:
        out = dec.unmask_tokens(tokens, masked, **args)
    finally:
        dec.reference_compat = False
    assert np.array_equal(out.cpu().numpy(), fix["out/tokens"])
    plain = dec.unmask_tokens(tokens, masked, **args)
    assert not np.array_equal(plain.cpu().numpy(), fix["out/tokens"])     # the default keeps the intended row


def test_render_session_serves_cross_attending_decoders(dev):
    """`decode.RenderSession` with context_emb_mode='attention' (inference/generators.py:230-240): the window's score embeddings are the
    attended context and grow with every call; keys / values are appended to static buffers and the captured step reads the context
    length from device memory.  Teacher-forced on the fixture's tokens, two notes per call, the session's predictions equal those of the
    module path called the reference's way (`unmask_tokens` with caches cut to the known prefix and the window's context) except at
    near-ties of the bf16 module path."""
    from scoreperformer_amd.decode import RenderSession
    from scoreperformer_amd.inference import ScorePerformerGenerator
    from scoreperformer_amd.modules.sampling import top_k
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy_xattn.npz"), allow_pickle=False))
    model, sd, cfg = _xattn_model(dev)
    dec = model.perf_decoder
    dims = [3, 5, 10, 11]
    truth = torch.from_numpy(fix["out/tokens"][0]).to(dev)                 # [40, 12]
    masked = torch.from_numpy(fix["in/masked_perf"][0]).to(dev)
    ctx = torch.from_numpy(fix["out/score_embeddings"][0]).to(dev)
    sty = torch.from_numpy(fix["out/perf_embeddings"][0]).to(dev)
    sess = RenderSession(dec.model, 64, dims)
    assert sess.cross
    dec.use_decode_engine = False    # the module path proper on the reference side (caches=None would otherwise go through the engine)
    caches, agree, total, g = None, 0, 0, 2
    for k in range(4, 38, g):
        Lin = k + g
        model_in = truth[:Lin].clone()
        model_in[k:Lin, dims] = 1
        doubled = masked[:Lin]
        sess.truncate(k - 1)
        got = sess.decode(model_in, doubled, ctx[:Lin], sty[:Lin], g).cpu().numpy()
        if caches is not None:
            caches = ScorePerformerGenerator.cut_caches(caches, right_idx=k - 1)
        want, caches = dec.unmask_tokens(model_in, doubled, context=ctx[:Lin].unsqueeze(0), style_embeddings=sty[:Lin].unsqueeze(0),
                                         caches=caches, return_caches=True, filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        want = want[-g:].cpu().numpy()
        assert (got[:, [d for d in range(12) if d not in dims]] == want[:, [d for d in range(12) if d not in dims]]).all()
        agree += int((got[:, dims] == want[:, dims]).sum()); total += g * len(dims)
    # the module path runs bf16 GEMMs: random-weight logits flip at near-ties (8 of 136 seen); the exact statement is the one below
    assert sess.steps_run > 30 and agree >= 0.90 * total, (agree, total)
    # EXACT: one call over the whole fixture window with the whole context = the CPU oracle's greedy tokens (fp32 on both sides), from a
    # session that has just served other windows (reset + static-buffer reuse)
    from oracle import ref_cpu
    tokens, maskedb = torch.from_numpy(fix["in/tokens"]), torch.from_numpy(fix["in/masked_perf"])
    want = ref_cpu.greedy_unmask(sd, cfg, tokens, maskedb, torch.from_numpy(fix["out/score_embeddings"]),
                                 torch.from_numpy(fix["out/perf_embeddings"]),
                                 context_mask=torch.ones(1, ctx.shape[0], dtype=torch.bool)).numpy()[0]   # every context row valid
    first = int((tokens[0] == 1).any(dim=1).nonzero().min())
    n = tokens.shape[1] - first
    sess.reset()
    got = sess.decode(tokens[0].to(dev), maskedb[0].to(dev), ctx, sty, n).cpu().numpy()
    assert (got == want[-n:]).all()


@pytest.mark.parametrize("tag", ["incl", "excl"])
def test_latent_dropout_matches_reference_golden(dev, tag):
    """base.yaml:119-126 trains with latent_dropout [0, .1, .2, .4]: the reference's own per-level drop masks (recorded from
    `dropout_latent_mask`, mmd_transformer.py:537-542) go through `_drop_override`; embeddings, the combined inclusive / exclusive
    mask incl. the dead-pan exemption (mmd_transformer.py:249-253,284-291), loss and every gradient norm must match the reference."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    fix = dict(np.load(os.path.join(GOLD, "latent_dropout.npz"), allow_pickle=False))
    kw = dict(preset="tiny", num_tokens=SMALL_VOCAB, latent_dropout=[0.0, 0.1, 0.2, 0.4])
    cfg = model_config(**kw)
    cfg.perf_encoder.inclusive_latent_dropout = tag == "incl"
    model = ScorePerformer.init(cfg)
    assert model.perf_encoder.inclusive_latent_dropout == (tag == "incl")
    model.load_state_dict(filled_state_dict(model, seed=0), strict=True)
    arena = ParamArena(model, dev)
    model.train()
    batch = {k[3:]: torch.from_numpy(v).to(dev) for k, v in fix.items() if k.startswith("in/")}
    model.perf_encoder._z_override = [torch.from_numpy(fix[f"{tag}/z/{i}"]).to(dev) for i in range(4)]
    model.perf_encoder._drop_override = [torch.from_numpy(fix[f"{tag}/drop/{i}"][..., 0]).to(dev) if f"{tag}/drop/{i}" in fix else None
                                         for i in range(4)]
    out = model(**batch)
    enc = out.perf_encoder
    np.testing.assert_array_equal(enc.dropout_mask.cpu().numpy(), fix[f"{tag}/dropout_mask"])
    full, emb = enc.full_embeddings.detach().float().cpu().numpy(), enc.embeddings.detach().float().cpu().numpy()
    assert np.abs(full - fix[f"{tag}/full_embeddings"]).max() <= 0.03 * np.abs(fix[f"{tag}/full_embeddings"]).max() + 1e-3
    assert np.abs(emb - fix[f"{tag}/embeddings"]).max() <= 0.03 * np.abs(fix[f"{tag}/embeddings"]).max() + 1e-3
    assert (emb[fix[f"{tag}/dropout_mask"]] == 0).all() and (emb != full).any()
    pre = f"{tag}/losses/"
    ref_losses = {k[len(pre):]: float(v) for k, v in fix.items() if k.startswith(pre)}
    assert set(out.losses) == set(ref_losses)
    for k, v in ref_losses.items():
        assert abs(float(out.losses[k]) - v) <= 2e-2 * max(1.0, abs(v)), (k, float(out.losses[k]), v)
    assert abs(float(out.loss) - float(fix[f"{tag}/loss"])) <= 2e-2 * float(fix[f"{tag}/loss"])
    hs = out.perf_decoder.hidden_state.detach().float().cpu().numpy()
    assert np.abs(hs - fix[f"{tag}/hidden_state"]).max() <= 0.03 * np.abs(fix[f"{tag}/hidden_state"]).max()
    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    bad, checked = [], 0
    for k, v in fix.items():
        if k.startswith(f"{tag}/gradnorm/"):
            got = float(named[k[len(tag) + 10:]].grad.double().norm())
            checked += 1
            if abs(got - float(v)) > 0.06 * float(v) + 2e-3:
                bad.append((k, got, float(v)))
        if k.startswith(f"{tag}/grad/"):
            g = named[k[len(tag) + 6:]].grad.float().cpu().numpy()
            assert np.abs(g - v).max() <= 0.06 * np.abs(v).max() + 1e-4, k
    assert checked > 100 and not bad, bad[:10]


def test_latent_dropout_own_draws(dev):
    """Without the override the product draws its own masks: whole latent vectors per segment, never on the `mean` level, never in a
    dead-pan sample, inclusive across levels (a note dropped on the bar level is dropped on beat and onset level too), and at the
    configured rates; eval mode drops nothing (mmd_transformer.py:284-291,351-362)."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    p = [0.0, 0.1, 0.2, 0.4]
    cfg = model_config("tiny", num_tokens=SMALL_VOCAB, latent_dropout=p)
    torch.manual_seed(3)
    model = ScorePerformer.init(cfg)
    ParamArena(model, dev)
    model.train()
    batch = {k: v.to(dev) for k, v in synthetic_batch(64, 128, num_tokens=SMALL_VOCAB, ragged=True, seed=9).items()}
    batch["deadpan_mask"][:8] = True
    out = model(**batch).perf_encoder
    dm, full, emb = out.dropout_mask, out.full_embeddings, out.embeddings
    L = list(cfg.perf_encoder.latent_dim)
    lv = [t[..., 0] for t in dm.split(L, dim=-1)]
    for t, piece in zip(lv, dm.split(L, dim=-1)):
        assert bool((piece == t[..., None]).all())             # whole vectors
    assert not bool(lv[0].any()) and not bool(dm[:8].any()) and not bool(dm[~batch["perf_mask"]].any())
    assert bool((lv[1] <= lv[2]).all()) and bool((lv[2] <= lv[3]).all())          # inclusive
    # constant within a segment (the bar level only: the synthetic beat / onset ids are drawn independently of the bars, so a beat may
    # straddle a bar line and the INCLUSIVE beat mask changes there)
    same = batch["bars"][:, 1:] == batch["bars"][:, :-1]
    assert bool((lv[1][:, 1:] == lv[1][:, :-1])[same & batch["perf_mask"][:, 1:]].all())
    assert bool((lv[1][:, 1:] != lv[1][:, :-1]).any())
    live = batch["perf_mask"][8:]
    want = [1 - (1 - p[1]), 1 - (1 - p[1]) * (1 - p[2]), 1 - (1 - p[1]) * (1 - p[2]) * (1 - p[3])]
    for t, w in zip(lv[1:], want):
        got = float(t[8:][live].float().mean())
        assert abs(got - w) < 0.05, (got, w)
    assert bool((emb == full * (~dm)).all())
    model.eval()
    with torch.no_grad():
        ev = model(**batch).perf_encoder
    assert ev.dropout_mask is None and ev.embeddings is ev.full_embeddings or bool((ev.embeddings == ev.full_embeddings).all())


def test_decode_engine_serves_a_batch_of_sequences_with_one_mask_layout(dev):
    """`unmask_tokens` with b > 1 (the reference's loop takes the MASK layout of batch element 0 for every sequence, wrappers.py:385-396):
    un-padded sequences with one layout go through the fp32 decode engine one after the other.  Row 0 is the reference's own fixture and
    must come out token for token; every row equals its own single-sequence engine call bit for bit (tokens AND caches, in the reference's
    batch-first layouts); the CPU oracle's greedy loop confirms a perturbed row; a RIGHT-padded batch takes the engine too (every sequence over
    its valid prefix) and so does a FRONT-padded one whose blocks start with a given note (round 6); a block whose first note is itself to be
    decoded, holes in the mask, or rows with different layouts still take the module path."""
    from oracle import ref_cpu
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd import decode
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config
    fix = dict(np.load(os.path.join(GOLD, "tiny_greedy.npz"), allow_pickle=False))
    cfg = model_config(preset="tiny", num_tokens=SMALL_VOCAB)
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=SMALL_VOCAB))
    sd = filled_state_dict(model, seed=3)
    model.load_state_dict(sd)
    ParamArena(model, dev)
    model.eval()
    tokens = torch.from_numpy(fix["in/tokens"])
    masked = torch.from_numpy(fix["in/masked_perf"])
    ctx = torch.from_numpy(fix["out/score_embeddings"])
    sty = torch.from_numpy(fix["out/perf_embeddings"])
    g = torch.Generator().manual_seed(9)
    # rows 1 and 2: the same MASK layout, other un-masked sub-tokens (score side) and other encoder outputs
    tok_b, msk_b = tokens.repeat(3, 1, 1), masked.repeat(3, 1, 1)
    for r in (1, 2):
        keep = tok_b[r] != 1
        noise = torch.randint(4, 10, tok_b[r].shape, generator=g)
        tok_b[r] = torch.where(keep & (tok_b[r] > 3), noise, tok_b[r])
        msk_b[r] = torch.where(msk_b[r] != 1, tok_b[r], msk_b[r])
    ctx_b = torch.cat([ctx, ctx + 0.05 * torch.randn(ctx.shape, generator=g), ctx.flip(1)], 0)
    sty_b = torch.cat([sty, sty * 0.9, sty + 0.05 * torch.randn(sty.shape, generator=g)], 0)
    dec = model.perf_decoder
    runs = []
    real_run = decode.GreedyDecoder.run

    def counting_run(self, *a, **k):
        runs.append(1)
        return real_run(self, *a, **k)

    decode.GreedyDecoder.run = counting_run
    try:
        kw = dict(filter_logits_fn=top_k, filter_kwargs={"k": 1}, return_caches=True, disable_tqdm=True)
        out_b, caches_b = dec.unmask_tokens(tok_b.to(dev), msk_b.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev), **kw)
        assert len(runs) == 3                                             # three engine runs, no module path
        assert int((out_b[0].cpu().numpy() != fix["out/tokens"][0]).sum()) == 0
        assert tuple(caches_b.token_emb.shape) == (3,) + tuple(fix["cache/token_emb_shape"])[1:]
        assert tuple(caches_b.transformer.attention[0].keys.shape) == (3,) + tuple(fix["cache/keys0_shape"])[1:]
        for r in range(3):
            out_r, caches_r = dec.unmask_tokens(tok_b[r:r + 1].to(dev), msk_b[r:r + 1].to(dev), context=ctx_b[r:r + 1].to(dev),
                                                style_embeddings=sty_b[r:r + 1].to(dev), **kw)
            assert torch.equal(out_r[0], out_b[r])
            assert torch.equal(caches_r.token_emb[0], caches_b.token_emb[r])
            for a, b_ in zip(caches_r.transformer.hiddens, caches_b.transformer.hiddens):
                assert torch.equal(a[0], b_[r])
            assert torch.equal(caches_r.transformer.attention[-1].values[0], caches_b.transformer.attention[-1].values[r])
        # the oracle's greedy loop on the perturbed row 1
        want = ref_cpu.greedy_unmask(sd, cfg, tok_b[1:2], msk_b[1:2], ctx_b[1:2], sty_b[1:2]).numpy()
        got = out_b[1:2].cpu().numpy()
        assert (got != want).sum() <= 0.03 * (want != tok_b[1:2].numpy()).sum()
        # a RIGHT-padded batch goes through the engine too, every sequence over its own valid prefix (a causal decoder's valid positions
        # never see the padded keys behind them); the padded tail comes back as it was given, caches zero-padded to the longest prefix
        n_before = len(runs)
        Lb = tok_b.shape[1]
        cut = Lb - 5
        pad_mask = torch.ones(tok_b.shape[:2], dtype=torch.bool)
        pad_mask[2, cut:] = False
        tok_p, msk_p = tok_b.clone(), msk_b.clone()
        tok_p[2, cut:], msk_p[2, cut:] = 0, 0                      # pad tokens behind the notes of row 2
        out_p, caches_p = dec.unmask_tokens(tok_p.to(dev), msk_p.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev),
                                            mask=pad_mask.to(dev), **kw)
        assert len(runs) == n_before + 3 and tuple(out_p.shape) == tuple(tok_b.shape)
        assert torch.equal(out_p[:2], out_b[:2])                     # the full rows: as without padding
        assert int((out_p[2, cut:].cpu() != 0).sum()) == 0           # the padded tail is left alone
        out_t, caches_t = dec.unmask_tokens(tok_p[2:3, :cut].to(dev), msk_p[2:3, :cut].to(dev), context=ctx_b[2:3, :cut].to(dev),
                                            style_embeddings=sty_b[2:3, :cut].to(dev), **kw)
        assert torch.equal(out_t[0], out_p[2, :cut])                 # row 2 = its own trimmed single-sequence call, bit for bit
        nt = caches_t.token_emb.shape[1]
        assert torch.equal(caches_t.token_emb[0], caches_p.token_emb[2, :nt]) and not caches_p.token_emb[2, nt:].any()
        assert torch.equal(caches_t.transformer.hiddens[-1][0], caches_p.transformer.hiddens[-1][2, :nt])
        kt, kp = caches_t.transformer.attention[0].keys, caches_p.transformer.attention[0].keys
        assert torch.equal(kt[0], kp[2, :nt] if kp.ndim == 3 else kp[2, :, :nt])
        # padding in FRONT of the notes (round 6): the engine runs every sequence over its own block of notes -- masked keys in front are
        # invisible and every position-dependent term is relative (ALiBi) or per note.  The block's first note must be given (it has no
        # valid predecessor to be predicted from), here in every row because the rows share one MASK layout.
        n_before = len(runs)
        front = 3
        f_mask = torch.ones(tok_b.shape[:2], dtype=torch.bool)
        f_mask[:, :front] = False
        f_mask[2, cut:] = False                                      # row 2: padded on both sides
        tok_f, msk_f = tok_b.clone(), msk_b.clone()
        tok_f[:, :front], msk_f[:, :front] = 0, 0
        tok_f[2, cut:], msk_f[2, cut:] = 0, 0
        tok_f[:, front] = out_b[:, front].cpu()                      # the block's first note: given in full (no MASK sub-token)
        msk_f[:, front] = tok_f[:, front]
        out_f, caches_f = dec.unmask_tokens(tok_f.to(dev), msk_f.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev),
                                            mask=f_mask.to(dev), **kw)
        assert len(runs) == n_before + 3 and tuple(out_f.shape) == tuple(tok_b.shape)
        assert int((out_f[:, :front].cpu() != 0).sum()) == 0 and int((out_f[2, cut:].cpu() != 0).sum()) == 0     # padding left alone
        assert int((out_f.cpu()[f_mask] == 1).sum()) == 0                                                         # every note decoded
        for r, end in ((0, Lb), (1, Lb), (2, cut)):
            out_t, caches_t = dec.unmask_tokens(tok_f[r:r + 1, front:end].to(dev), msk_f[r:r + 1, front:end].to(dev),
                                                context=ctx_b[r:r + 1, front:end].to(dev), style_embeddings=sty_b[r:r + 1, front:end].to(dev), **kw)
            assert torch.equal(out_t[0], out_f[r, front:end])        # = the trimmed single-sequence call, bit for bit
            nt = caches_t.token_emb.shape[1]
            assert torch.equal(caches_t.token_emb[0], caches_f.token_emb[r, front:front + nt]) and not caches_f.token_emb[r, :front].any()
            kt, kf = caches_t.transformer.attention[0].keys, caches_f.transformer.attention[0].keys
            assert torch.equal(kt[0], kf[r, front:front + nt] if kf.ndim == 3 else kf[r, :, front:front + nt])
        # and the module path (concatenated caches, bf16 GEMMs, the full mask) decodes the same notes up to bf16 near-ties
        dec.use_decode_engine = False
        out_m = dec.unmask_tokens(tok_f.to(dev), msk_f.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev), mask=f_mask.to(dev),
                                  filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        dec.use_decode_engine = True
        valid = f_mask[..., None].expand_as(out_f).to(dev)
        assert float((out_m[valid] == out_f[valid]).float().mean()) > 0.97
        # a first note that is itself to be decoded, or holes in the mask, keep the module path
        n_before = len(runs)
        odd_mask = torch.ones(tok_b.shape[:2], dtype=torch.bool)
        odd_mask[1, :3] = False
        out_o = dec.unmask_tokens(tok_b.to(dev), msk_b.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev), mask=odd_mask.to(dev),
                                  filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        assert len(runs) == n_before and tuple(out_o.shape) == tuple(tok_b.shape)
        hole_mask = torch.ones(tok_b.shape[:2], dtype=torch.bool)
        hole_mask[0, 7] = False
        dec.unmask_tokens(tok_b.to(dev), msk_b.to(dev), context=ctx_b.to(dev), style_embeddings=sty_b.to(dev), mask=hole_mask.to(dev),
                          filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        assert len(runs) == n_before
    finally:
        decode.GreedyDecoder.run = real_run
