"""GPU: loss AND gradient parity at C2 scale (BASELINE config 2's model: d=512, 8 heads MQA, 6/6/6 layers, GLU-SiLU x4, MMD-VAE,
tied head, 71.9 M parameters at their own initialisation; 2 sequences x 1024 notes, one ragged, one dead-pan; training mode,
dropout 0) against the fp32 CPU oracle on identical inputs.

north_star's tolerance: |loss_HIP - loss_CPU| <= 1e-3 (bf16 GEMM operands; fp32 accumulation, softmax, statistics, residual stream).
Measured on MI355X (tools/parity_c2.py, profiles/r02_parity_c2.txt): 4.5e-4, every loss-dict entry within 4.5e-4.

Gradients, per tensor: ||g_HIP - g_CPU|| <= 0.06 ||g_CPU|| + 5e-4.  Measured relative L2 errors: 0.8-1.3 % in the style encoder and
the VAE heads, 3.3-4.1 % where the signal has crossed twelve bf16 layers (score encoder, decoder), median 3.2 % over all 314 tensors.
The absolute term is the bf16 noise floor of sums that cancel: at initialisation attention is near-uniform and the q / k projection
gradients are ~1e-4 in norm next to ~0.9 for the v projections of the same layers, so their errors (3e-5 .. 3e-4 absolute) are of the
size of the signal itself.  The rule is asserted for 17 named tensors AND for every one of the 314 parameter tensors.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REL, FLOOR = 0.06, 5e-4

# named in the assertion message when they fail; the same rule covers all other tensors
NAMED = [
    "score_encoder.transformer.layers.0.1.to_q.weight",
    "score_encoder.transformer.layers.11.1.ff.3.weight",
    "score_encoder.token_emb.project_emb.weight",
    "perf_encoder.transformer.layers.4.1.to_v.weight",
    "perf_encoder.transformer.layers.5.1.ff.0.proj.weight",
    "perf_encoder.transformer.final_norm.weight",
    "perf_encoder.vae_head.bar_mean.linear.weight",
    "perf_decoder.model.transformer.layers.0.0.0.linear.weight",
    "perf_decoder.model.transformer.layers.4.1.to_out.weight",
    "perf_decoder.model.transformer.layers.7.1.ff.0.proj.bias",
    "perf_decoder.model.transformer.layers.10.1.to_k.weight",
    "perf_decoder.model.token_emb.project_multiemb.weight",
    "perf_decoder.model.token_emb.project_emb.weight",                   # tied with the LM head's projection
    "perf_decoder.model.project_emb.weight",
    "perf_decoder.model.lm_head.norm.weight",
    "score_encoder.token_emb.embs.Velocity.value_layer.1.0.weight",      # tied: the tables of all three transformers + LM head
    "perf_decoder.model.transformer.layers.4.1.rel_pos.learned_logslopes",
]
GRAD_BOUNDS = {k: REL for k in NAMED}


def run_c2(dev, threads=32, preset="c2", seq=1024, seed=21):
    from oracle import ref_cpu
    from oracle.weights import canonical
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    cfg = model_config(preset)
    torch.manual_seed(1234)
    model = ScorePerformer.init(model_config(preset))    # the model's own initialisation (what a training run starts from)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    arena = ParamArena(model, dev)
    model.train()
    batch = synthetic_batch(2, seq, seed=seed, ragged=True, deadpan_p=0.25)
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(100 + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    model.perf_encoder._z_override = [t.to(dev) for t in z]
    out = model(**{k: v.to(dev) for k, v in batch.items()})
    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()

    torch.set_num_threads(max(1, min(os.cpu_count() or 1, threads)))
    leaves, sdg = {}, {}
    for k, v in sd.items():   # tied tensors share one leaf so that gradients accumulate like in the module tree
        leaf = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v
        sdg[k] = leaves.setdefault(canonical(k), leaf)
    ref = ref_cpu.score_performer_forward(sdg, cfg, batch, z, training=True)
    ref["loss"].backward()
    return model, out, ref, sdg


def grad_errors(model, sdg, names):
    named = dict(model.named_parameters())
    rows = []
    for k in names:
        g, r = named[k].grad.detach().float().cpu(), sdg[k].grad
        rows.append((k, float((g - r).norm() / r.norm().clamp_min(1e-30)), float(r.norm())))
    return rows


def all_grad_names(model, sdg):
    seen, names = set(), []
    for k, p in model.named_parameters():
        if id(p) not in seen and sdg[k].grad is not None:
            seen.add(id(p)); names.append(k)
    return names


def test_c2_loss_and_gradients_match_the_cpu_oracle(dev):
    model, out, ref, sdg = run_c2(dev)
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    assert abs(got - want) <= 1e-3, (got, want)
    for k, v in ref["losses"].items():
        assert abs(float(out.losses[k]) - float(v.detach())) <= 1e-3, (k, float(out.losses[k]), float(v.detach()))
    names = all_grad_names(model, sdg)
    assert len(names) >= 300 and all(k in names for k in NAMED)
    rows = grad_errors(model, sdg, names)
    bad = [(k, e * n, n) for k, e, n in rows if not e * n <= REL * n + FLOOR]   # (tensor, absolute L2 error, reference norm)
    assert not bad, ([b for b in bad if b[0] in NAMED], bad[:10])
    # and the gradient as one vector: relative L2 error <= 2.5 % (measured 1.32 %, tools/parity_measure.py, round 6), norm within 1 %
    err2 = sum((e * n) ** 2 for _, e, n in rows) ** 0.5
    ref2 = sum(n ** 2 for _, _, n in rows) ** 0.5
    assert err2 <= 0.025 * ref2, (err2, ref2)
    named = dict(model.named_parameters())
    got2 = sum(float(named[k].grad.double().pow(2).sum()) for k in names) ** 0.5
    assert abs(got2 - ref2) <= 1e-2 * ref2, (got2, ref2)


def test_c3_step_matches_oracle(dev):
    """BASELINE config 3's workload as a test (the bench line's parity leg, moved into pytest): the C3 model (max_seq_len 2048),
    2 sequences x 2048 notes, one ragged, training mode, dropout 0, forward + backward through the HIP path against the fp32 CPU oracle
    on identical inputs, weights and N(0, I) samples.  |loss_HIP - loss_CPU| <= 1e-3 (north_star), every loss-dict entry within 1e-3,
    the whole gradient as one vector within 2.5 % relative L2 (measured 1.40 %) and 1 % in norm, and -- since round 6 -- EVERY one of the
    314 gradient tensors within 7 % + 5e-4 of its reference (measured: median 3.5 %, at most 4.0 % for tensors of norm > 1e-2; twice the
    depth in tokens of the C2 test, whose 6 % rule the worst tensor here meets at 0.96 of the bound)."""
    model, out, ref, sdg = run_c2(dev, preset="c3", seq=2048, seed=33)
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    assert abs(got - want) <= 1e-3, (got, want)
    for k, v in ref["losses"].items():
        assert abs(float(out.losses[k]) - float(v.detach())) <= 1e-3, (k, float(out.losses[k]), float(v.detach()))
    names = all_grad_names(model, sdg)
    assert len(names) >= 300
    rows = grad_errors(model, sdg, names)
    bad = [(k, e * n, n) for k, e, n in rows if not e * n <= 0.07 * n + FLOOR]
    assert not bad, bad[:10]
    err2 = sum((e * n) ** 2 for _, e, n in rows) ** 0.5
    ref2 = sum(n ** 2 for _, _, n in rows) ** 0.5
    assert err2 <= 0.025 * ref2, (err2, ref2)
    named = dict(model.named_parameters())
    got2 = sum(float(named[k].grad.double().pow(2).sum()) for k in names) ** 0.5
    assert abs(got2 - ref2) <= 1e-2 * ref2, (got2, ref2)


# ---------------------------------------------------------------------------------------------------------------------------
# Loss parity WITH dropout (the benchmarked configuration: attention + feed-forward dropout 0.1, latent dropout [0, .1, .2, .4]).
# The product's masks are counter-based (they cannot equal torch's draws), so the product's OWN masks go into the oracle: every
# dropout site of the product's forward is logged (module, seed, shape), its mask is read back through the same kernels on crafted
# inputs (attention: q = k = 0 and one-hot value blocks make o = keep / (nk (1 - p)); feed-forward: value 1, gate 20), and
# `oracle.ref_cpu.DROP_FEED` applies keep / (1 - p) at attend.py:122 / feedforward.py:57-60.  What the comparison then covers is
# everything the statistical dropout tests cannot: that the masked probabilities / activations enter P V, the output projection and
# every gradient the way the reference's F.dropout does.
# ---------------------------------------------------------------------------------------------------------------------------

class _DropLog:
    def __init__(self, model, monkeypatch):
        from scoreperformer_amd import ops
        from scoreperformer_amd.modules.transformer.attention import Attention
        from scoreperformer_amd.modules.transformer.feedforward import FeedForward
        self.sites, self.stack = [], []
        for name, mod in model.named_modules():
            if isinstance(mod, (Attention, FeedForward)):
                mod.register_forward_pre_hook(lambda m, a, name=name: self.stack.append(name))
                mod.register_forward_hook(lambda m, a, o: (self.stack.pop(), None)[1])   # (a hook's return value would replace the output)
        raw_attn, raw_glu, raw_act = ops.attn_fwd, ops.gemm_glu, ops.act_fwd

        def attn_fwd(q, k, v, **kw):
            if kw.get("p_drop", 0.0) > 0:
                self.sites.append(("attn", self.stack[-1], kw["seed"], kw["p_drop"], (q.shape[0], q.shape[2], q.shape[1], k.shape[1])))
            return raw_attn(q, k, v, **kw)

        def gemm_glu(x, w, bias, **kw):
            if kw.get("p_drop", 0.0) > 0:
                self.sites.append(("ffn", self.stack[-1], kw["seed"], kw["p_drop"], (x.shape[0], w.shape[0] // 2)))
            return raw_glu(x, w, bias, **kw)

        def act_fwd(u, **kw):   # (row counts the fused projection does not take: GEMM + activation kernel, same mask function)
            if kw.get("p_drop", 0.0) > 0:
                self.sites.append(("ffn", self.stack[-1], kw["seed"], kw["p_drop"], (u.numel() // u.shape[-1], u.shape[-1] // 2)))
            return raw_act(u, **kw)

        monkeypatch.setattr(ops, "attn_fwd", attn_fwd)
        monkeypatch.setattr(ops, "gemm_glu", gemm_glu)
        monkeypatch.setattr(ops, "act_fwd", act_fwd)
        self.raw_attn, self.raw_act = raw_attn, raw_act

    def masks(self, dev):
        """{(kind, state_dict prefix): multiplier keep / (1 - p) on the CPU}, read back through the product's kernels."""
        from scoreperformer_amd import ops
        out = {}
        for kind, name, seed, p, shape in self.sites:
            if kind == "attn":
                b, h, nq, nk = shape
                q = torch.zeros(b, nq, h, 64, device=dev, dtype=torch.bfloat16)
                k = torch.zeros(b, nk, 1, 64, device=dev, dtype=torch.bfloat16)
                keep = torch.empty(b, h, nq, nk, dtype=torch.bool)
                for j0 in range(0, nk, 64):
                    w = min(64, nk - j0)
                    v = torch.zeros(b, nk, 1, 64, device=dev, dtype=torch.bfloat16)
                    v[:, j0:j0 + w, 0, :w] = torch.eye(w, device=dev, dtype=torch.bfloat16)
                    o = self.raw_attn(q, k, v, p_drop=p, seed=seed)[0]                      # [b, nq, h, 64] = keep / (nk (1 - p))
                    keep[..., j0:j0 + w] = (o[..., :w].float() * nk > 0.5).permute(0, 2, 1, 3).cpu()
            else:
                M, I = shape
                u = torch.ones(M, 2 * I, device=dev, dtype=torch.bfloat16)
                u[:, I:] = 20.0
                keep = (self.raw_act(u, act=0, glu=True, p_drop=p, seed=seed).float() > 1.0).cpu()
            assert abs(float(keep.float().mean()) - (1 - p)) < 5e-3, (kind, name, float(keep.float().mean()))
            out[(kind, name + ".")] = keep.float() / (1.0 - p)
        return out


def test_c2_loss_with_dropout_matches_the_oracle_fed_with_the_products_masks(dev, monkeypatch):
    from oracle import ref_cpu
    from oracle.weights import canonical
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    kw = dict(preset="c2", dropout=0.1, latent_dropout=[0.0, 0.1, 0.2, 0.4])
    cfg = model_config(**kw)
    torch.manual_seed(1234)
    model = ScorePerformer.init(model_config(**kw))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    arena = ParamArena(model, dev)
    model.train()
    seq = 1024
    batch = synthetic_batch(2, seq, seed=21, ragged=True, deadpan_p=0.0)
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(100 + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    gen = torch.Generator().manual_seed(77)
    lat = [None] + [torch.rand(2, int(batch[k].max()) + 1, generator=gen) < p for k, p in (("bars", .1), ("beats", .2), ("onsets", .4))]
    model.perf_encoder._z_override = [t.to(dev) for t in z]
    model.perf_encoder._drop_override = [None if m is None else m.to(dev) for m in lat]
    log = _DropLog(model, monkeypatch)
    out = model(**{k: v.to(dev) for k, v in batch.items()})
    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    assert len([s for s in log.sites if s[0] == "attn"]) == 18 and len([s for s in log.sites if s[0] == "ffn"]) == 18
    mult = log.masks(dev)
    used = set()

    def feed(kind, prefix, t):
        used.add((kind, prefix))
        m = mult[(kind, prefix)]
        return t * (m if kind == "attn" else m.view(t.shape))

    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    leaves, sdg = {}, {}
    for k, v in sd.items():
        leaf = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v
        sdg[k] = leaves.setdefault(canonical(k), leaf)
    monkeypatch.setattr(ref_cpu, "DROP_FEED", feed)
    ref = ref_cpu.score_performer_forward(sdg, cfg, batch, z, training=True, drop_masks=[None if m is None else m[..., None] for m in lat])
    ref["loss"].backward()
    assert used == set(mult)                                   # every logged site was applied by the oracle, under the same module path
    got, want = float(out.loss.detach()), float(ref["loss"].detach())
    assert abs(got - want) <= 1e-3, (got, want)
    for k, v in ref["losses"].items():
        assert abs(float(out.losses[k].detach()) - float(v.detach())) <= 1e-3, (k, float(out.losses[k].detach()), float(v.detach()))
    names = all_grad_names(model, sdg)
    rows = grad_errors(model, sdg, names)
    bad = [(k, e * n, n) for k, e, n in rows if not e * n <= REL * n + FLOOR]      # the per-tensor rule of the dropout-free test
    assert not bad, bad[:10]
    err2 = sum((e * n) ** 2 for _, e, n in rows) ** 0.5
    ref2 = sum(n ** 2 for _, _, n in rows) ** 0.5
    assert err2 <= 0.05 * ref2, (err2, ref2)
    named = dict(model.named_parameters())
    got2 = sum(float(named[k].grad.double().pow(2).sum()) for k in names) ** 0.5
    assert abs(got2 - ref2) <= 1e-2 * ref2, (got2, ref2)
    # How much the masks matter.  At initialisation the LOSS hardly feels them (5.1103 without vs 5.1089 with: the branch outputs are small
    # next to the residual stream) and weight gradients average them out over 2048 tokens; the ACTIVATIONS feel them token by token: the
    # same oracle WITHOUT the attention / feed-forward masks is 8-12 % away from the masked one in the final hidden states of the score
    # encoder and of the decoder, the HIP path 0.4-0.5 % (valid rows).
    monkeypatch.setattr(ref_cpu, "DROP_FEED", None)
    with torch.no_grad():
        plain = ref_cpu.score_performer_forward(sdg, cfg, batch, z, training=True, drop_masks=[None if m is None else m[..., None] for m in lat])
    report = []
    vs, vd = batch["score_mask"], batch["perf_mask"][:, :-1]
    for tag, h_hip, h_ref, h_plain, valid in (("score encoder", out.score_encoder.hidden_state, ref["score_embeddings"], plain["score_embeddings"], vs),
                                               ("decoder", out.perf_decoder.hidden_state, ref["hidden_state"], plain["hidden_state"], vd)):
        a, r, q = h_hip.detach().float().cpu()[valid], h_ref.detach()[valid], h_plain.detach()[valid]
        e, d = float((a - r).norm() / r.norm()), float((q - r).norm() / r.norm())
        report.append((tag, e, d))
        assert e <= 0.02 and d >= 3.0 * e, (tag, e, d)
    print(f"dropout parity: loss HIP {got:.6f} CPU {want:.6f} |d| {abs(got - want):.2e} (CPU without the masks {float(plain['loss']):.6f}); "
          f"gradient rel L2 error {err2 / ref2:.4f}; " + "; ".join(f"{t} hidden states: HIP error {e:.4f}, the unmasked oracle is {d:.4f} away" for t, e, d in report))
