"""Same-box A/B of the C5 greedy decode: five graph launches per decoder layer pair (SPN_DEC_PAIR=0) vs one persistent launch per pair
(csrc/decode_layer.hip, SPN_DEC_PAIR=<workgroups>).  Tokens and the final hidden rows must be identical bit for bit.
    python tools/bench_dec_pair.py [L]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def run(L, groups, dev, model, batch, enc):
    from scoreperformer_amd.modules.sampling import top_k
    os.environ["SPN_DEC_PAIR"] = str(groups)
    dec = model.perf_decoder
    # (unmask_tokens builds a fresh GreedyDecoder per call: the environment variable is read there)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, caches = dec.unmask_tokens(tokens, batch["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
                                        filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True, return_caches=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / (L - 1) * 1e6, out.cpu(), caches.transformer.hiddens[-1].float().cpu()


def timeline(L, groups, dev, model, batch, enc):
    """Per-phase time stamps (100 MHz constant clock) of the last note's launches: when each workgroup finished each phase."""
    from scoreperformer_amd.decode import GreedyDecoder
    os.environ["SPN_DEC_PAIR"], os.environ["SPN_DEC_PAIR_STAMPS"] = "1", "1"
    eng = GreedyDecoder(model.perf_decoder.model, L)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    eng.run(tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
    torch.cuda.synchronize()
    os.environ["SPN_DEC_PAIR_STAMPS"] = "0"
    groups = eng.pair_groups
    nA = eng.heads * eng.attn_splits
    nB = (eng.dim + 15) // 16
    roles = {"A (q|k|v rows, attention)": (slice(0, nA), {1: "qkv out", 3: "q gathered", 4: "keys done", 2: "partials out", 5: "tail: e out"}),
             "B (merge, projections)": (slice(nA, nA + nB), {7: "front: x in out", 3: "merged o out", 4: "x1 out", 6: "x out"}),
             "C (gated rows)": (slice(nA + nB, groups), {1: "weights requested", 2: "x1 gathered", 3: "norm done", 4: "rows done", 5: "g out",
                                                         6: "head: slab maxima out", 7: "head: token written"})}
    t_prev_end = None
    t_launch = None
    st0 = eng.pair_stamps[0].view(groups, 8).cpu().double() * 0.01
    entry = torch.cat([st0[:nA, 6], st0[nA:nA + nB, 5], st0[nA + nB:, 7]])
    first = float(st0[:, 0].min())
    print(f"kernel entry of the workgroups: {float(entry.min()) - first:6.2f} .. {float(entry.max()) - first:6.2f} us relative to the launch's first phase stamp")
    if len(eng.pair_stamps) > 1:   # the note's first phases (embed on the attention workgroups, front on the projection workgroups) stamp pair 1's record
        st1 = eng.pair_stamps[1].view(groups, 8).cpu().double() * 0.01
        for name, sl, k in (("embed: position known", slice(0, nA), 5), ("embed: table rows gathered", slice(0, nA), 6), ("embed: rows out", slice(0, nA), 7),
                            ("front: embedded tokens gathered", slice(nA, nA + nB), 5), ("front: x0 out", slice(nA, nA + nB), 7)):
            col = st1[sl, k]
            col = col[col > 0]
            if len(col):
                print(f"    {name}: {float(col.min()) - first:6.2f} .. {float(col.max()) - first:6.2f}")
    for pi, st in enumerate(eng.pair_stamps):
        st = st.view(groups, 8).cpu().double() * 0.01          # us
        t0 = float(st[:, 0].min())
        t_launch = t0 if t_launch is None else t_launch
        head = f"pair {pi} (t0 = {t0 - t_launch:6.2f} us after the launch's first stamp): workgroups start within {float(st[:, 0].max()) - t0:4.2f} us"
        if t_prev_end is not None:
            head += f", {t0 - t_prev_end:4.2f} us after the previous pair's last store"
        print(head)
        for role, (sl, cols) in roles.items():
            line = [f"    {role}:"]
            for k, name in cols.items():
                if pi == 1 and len(eng.pair_stamps) > 2 and ((role.startswith("A") and k in (5, 6, 7)) or (role.startswith("B") and k in (5, 7))):
                    continue   # the note's first phases (printed above)
                if pi == 0 and ((role.startswith("C") and k == 7) or (role.startswith("A") and k == 6) or (role.startswith("B") and k == 5)):
                    continue   # kernel-entry stamps (printed above)
                col = st[sl, k]
                col = col[col > 0]
                if len(col):
                    line.append(f"{name} {float(col.min()) - t0:5.2f}..{float(col.max()) - t0:5.2f}")
            print(" | ".join(line), flush=True)
        t_prev_end = float(st[nA:nA + nB, 6].max())


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    groups = [1]
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = ScorePerformer.init(model_config("c5", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=7, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    if os.environ.get("TIMELINE_ONLY") == "1":
        timeline(L, groups[0], dev, model, batch, enc)
        return
    us0, tok0, hid0 = run(L, 0, dev, model, batch, enc)
    print(f"L={L}  five launches per pair: {us0:.1f} us per note", flush=True)
    for g in groups:
        us, tok, hid = run(L, g, dev, model, batch, enc)
        same_t = bool((tok == tok0).all())
        same_h = bool(torch.equal(hid, hid0))
        print(f"L={L}  one persistent launch per pair: {us:.1f} us per note  tokens identical {same_t}  final hiddens bit-identical {same_h}"
              f"  (max |dh| {float((hid - hid0).abs().max()):.3g})", flush=True)
        us0b, _, _ = run(L, 0, dev, model, batch, enc)
        print(f"L={L}  five launches per pair (again): {us0b:.1f} us per note", flush=True)
    if os.environ.get("TIMELINE", "1") != "0":
        # the stamps are compiled into a variant library only (csrc/decode_layer.hip, -DSPN_DEC_STAMPS):
        #   python tools/build_variant.py decode_layer.hip stamps_spn.so -DSPN_DEC_STAMPS
        from scoreperformer_amd import lib as spn_lib
        if os.path.basename(spn_lib.LIB_PATH) != "stamps_spn.so":
            variant = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "stamps_spn.so")
            if not os.path.exists(variant):
                print("no tools/_bin/stamps_spn.so: build it with tools/build_variant.py (see the comment above) for the per-phase timeline")
                return
            import subprocess
            env = dict(os.environ, SPN_LIB=variant, TIMELINE_ONLY="1")
            sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), str(L)], env=env).returncode)
        timeline(L, groups[0], dev, model, batch, enc)


if __name__ == "__main__":
    main()
