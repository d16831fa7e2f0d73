"""GPU: BASELINE config 5 as a test -- the greedy performance render of ONE 4096-note sequence through the hipGraph-replayed decode
engine (C5 model: d=512, 8-head MQA, 6/6/6 layers, its own initialisation), tokens checked at EVERY decoded position.

The reference's per-note loop (`ScorePerformerMixedLMWrapper.unmask_tokens`, models/scoreperformer/wrappers.py:325-407) takes the
arg-max of the decoder's logits at position idx - 1 with PAD / MASK banned (wrappers.py:368-369).  Re-running the oracle's cache-free
loop over 4095 prefixes is quadratic on the CPU; instead the oracle (`oracle.ref_cpu.tuple_transformer`, fp32) is TEACHER-FORCED on the
engine's own tokens in one causal pass, which gives the reference logits of every position under exactly the prefix the engine saw.
Every engine token must be the oracle's arg-max; a differing token is accepted only as a near-tie below the fp32 margin (the engine's
token is the oracle's second choice and the oracle's top-2 logits differ by less than 5e-5 of the largest logit) -- the rule of
test_greedy_render_matches_reference_tokens, tightened because both sides compute in fp32 here.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
# measured on MI355X (profiles/r05_c5_decode_flips.json): the number of decoded tokens (of 16 380) that are not the fp32 oracle's arg-max.
# Every one of them must be a near-tie (the oracle's second choice within 5e-5 of the largest logit); the bound is the observed count + a
# margin of 2 for other boxes' rounding, not the 0.2 % the test accepted until round 4.
MAX_FLIPS = 2


def test_c5_decode_4096(dev):
    from oracle import ref_cpu
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch
    L = 4096
    torch.manual_seed(0)
    cfg = model_config("c5", max_seq_len=L)
    model = ScorePerformer.init(model_config("c5", max_seq_len=L))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=7)
    gb = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        enc = model.forward_encoders(perf=gb["perf"], perf_mask=gb["perf_mask"], score=gb["score"], score_mask=gb["score_mask"],
                                     bars=gb["bars"], beats=gb["beats"], onsets=gb["onsets"], deadpan_mask=gb["deadpan_mask"],
                                     compute_loss=False)
        tokens = gb["masked_perf"].clone()
        tokens[:, 0] = gb["perf"][:, 0]
        out = model.perf_decoder.unmask_tokens(tokens, gb["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
                                               filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
    torch.cuda.synchronize()
    got = out.cpu()
    assert int((got == 1).sum()) == 0, "MASK tokens left"
    assert tuple(got.shape) == (1, L, 12)
    # dims the loop does not predict are copied through
    keep = [d for d in range(12) if d not in PREDICTED_DIMS]
    assert torch.equal(got[..., keep], batch["masked_perf"][..., keep])

    # oracle, teacher-forced on the engine's tokens; the encoders' outputs are fed as the GPU produced them (decode path under test)
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    ctx = enc.score_embeddings.float().cpu()[:, 1:]
    sty = enc.perf_embeddings.float().cpu()[:, 1:]
    with torch.no_grad():
        _, logits = ref_cpu.tuple_transformer(sd, "perf_decoder.model.", cfg["perf_decoder"], [got[:, :-1], batch["masked_perf"][:, 1:]],
                                              causal=True, mask=torch.ones(1, L - 1, dtype=torch.bool), context=ctx, style=sty,
                                              with_logits=True)
    keys = list(logits.keys())
    checked, flips = 0, []
    for d in PREDICTED_DIMS:
        lg = logits[keys[d]][0].clone()            # [L - 1, V_d]: row t predicts position t + 1
        lg[:, :2] = -float("inf")                  # PAD / MASK banned (wrappers.py:368-369)
        want = lg.argmax(-1)
        have = got[0, 1:, d]
        masked = batch["masked_perf"][0, 1:, d] == 1
        checked += int(masked.sum())
        for t in torch.nonzero(masked & (want != have)).flatten().tolist():
            top2 = torch.topk(lg[t], 2)
            margin = float(top2.values[0] - top2.values[1])
            scale = float(lg[t][lg[t] > -1e30].abs().max())
            flips.append((d, t + 1, margin, scale, int(have[t]) == int(top2.indices[1])))
    assert checked >= 4 * (L - 2)
    # "bit-exact" as a NUMBER: how many of the checked tokens differ from the oracle's arg-max, printed (pytest -s / -rP shows it) and,
    # when SPN_PROFILE_DIR is set, stored next to the other measurements
    record = {"config": "C5 greedy render, 4096 notes, batch 1, decode engine", "tokens_checked": checked, "flips": len(flips),
              "flips_detail": [{"dim": d, "position": t, "top2_margin": m, "largest_logit": sc, "engine_token_is_second_choice": sec}
                               for d, t, m, sc, sec in flips]}
    print(f"C5 decode: {len(flips)} of {checked} decoded tokens differ from the fp32 oracle's arg-max")
    if os.environ.get("SPN_PROFILE_DIR"):
        import json
        with open(os.path.join(os.environ["SPN_PROFILE_DIR"], "c5_decode_flips.json"), "w") as fh:
            json.dump(record, fh, indent=1)
    bad = [f for f in flips if not (f[4] and f[2] <= 5e-5 * f[3] + 1e-6)]
    assert not bad, (len(flips), bad[:10])
    assert len(flips) <= MAX_FLIPS, (len(flips), checked, flips[:10])
