"""GPU: the persistent layer-pair launch of the decode engine (csrc/decode_layer.hip: one launch for the whole chain of pre-norm
('a', 'f') decoder layer pairs of a note, every phase handed over inside the launch through {epoch, value} granules) against the five launches it replaces.

Same arithmetic, association order and rounding by construction (both files compile without floating-point contraction), so the
comparison is EXACT: greedy tokens, every cached hidden row, every cached key / value row.  The five-launch engine itself is pinned to
the reference's tokens and to the CPU oracle elsewhere (tests/test_model_gpu.py, tests/test_c5_decode_gpu.py: those tests run the pair
launch, which is the default).  Reference path: models/scoreperformer/wrappers.py:325-407 -> modules/transformer/transformer.py:159-221.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(dev, preset, L, monkeypatch, attn_splits=16, mask_all=False, **cfg_kw):
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import GreedyDecoder
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    torch.manual_seed(3)
    model = ScorePerformer.init(model_config(preset, max_seq_len=L, **cfg_kw))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=11, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    if mask_all:            # every key of every note after the first is decoded (12 keys instead of the recipe's 4)
        batch["masked_perf"][:, 1:, :] = 1
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR", flag)
        eng = GreedyDecoder(model.perf_decoder.model, L, attn_splits=attn_splits)
        toks, n = eng.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
        torch.cuda.synchronize()
        out.append((eng, toks.clone(), n))
    return out


@pytest.mark.parametrize("preset,L,kw", [("tiny", 200, {}), ("tiny", 96, {"one_kv_head": False}), ("tiny", 96, {"style_emb_mode": "cat"}),
                                          ("tiny", 96, {"alibi_learned": False}), ("c5", 700, {})])
def test_pair_launch_equals_the_five_launches_bit_for_bit(dev, monkeypatch, preset, L, kw):
    (e0, t0, n0), (e1, t1, n1) = _engines(dev, preset, L, monkeypatch, **kw)
    assert e0.pair_groups == 0 and e1.pair_groups > 0, (e0.pair_groups, e1.pair_groups)
    chains = e1._pair_chains()
    assert len(chains) == 1 and sum(c.n for c in chains.values()) == len(e1.kc)   # ALL layer pairs of a note run as one launch
    assert n0 == n1 == L - 1 and int(e1.pair_err.item()) == 0
    assert e1.pair_front and e1.pair_tail      # the note's input projections and the LM head's input projection ride in the same launch
    assert e1.pair_head and e1.pair_embed      # ... and so do the arg-max LM head (greedy) and the token embeddings: ONE launch per note
    assert torch.equal(t0, t1)
    for a, b in zip(e0.hid + e0.kc + e0.vc, e1.hid + e1.kc + e1.vc):
        assert torch.equal(a[:n1], b[:n1])
    assert int(e1.pair_tick.item()) == n1                          # one epoch per decoded note


def test_pair_launch_serves_a_render_session_that_revisits_positions(dev, monkeypatch):
    """A RenderSession truncates and re-decodes positions (the generator's sliding window): the epoch counter keeps counting, so the tags of
    a re-decoded position never equal those left by its first visit.  Tokens equal the five-launch session's."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import RenderSession
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch
    L = 120
    torch.manual_seed(5)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=13, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    truth, masked = batch["perf"][0], batch["masked_perf"][0]
    ctx, sty = enc.score_embeddings[0], enc.perf_embeddings[0]
    dims = list(PREDICTED_DIMS)
    got = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR", flag)
        sess = RenderSession(model.perf_decoder.model, L, dims)
        rows = []
        for k in (20, 36, 28, 60, 44, 90):          # windows end here; going back re-decodes positions already visited
            g = 8
            win = truth[:k + g].clone()
            win[k:k + g, dims] = 1
            sess.truncate(min(sess.length, k - 1))
            rows.append(sess.decode(win, masked[:k + g], ctx[:k + g], sty[:k + g], g).clone())
        torch.cuda.synchronize()
        assert (sess.pair_groups > 0) == (flag == "1")
        got.append(torch.cat(rows))
    assert torch.equal(got[0], got[1])


def test_a_timed_out_hand_off_falls_back_to_the_five_launches(dev, monkeypatch):
    """`*err` != 0 (a hand-off poll ran into its bound: every poll loop of the persistent launch is bounded, later launches return at once)
    neither hangs the device nor breaks the engine: `run` clears the error word, drops the persistent launch for this engine, warns once
    and decodes the same window through the five launches per pair -- same tokens, same cache rows."""
    import time
    (e0, t0, n0), _ = _engines(dev, "tiny", 64, monkeypatch)
    from scoreperformer_amd.decode import GreedyDecoder
    from scoreperformer_amd.synthetic import synthetic_batch
    model_dec = e0.m
    monkeypatch.setenv("SPN_DEC_PAIR", "1")
    eng = GreedyDecoder(model_dec, 64)
    eng._pair_fault_inject = True
    batch = synthetic_batch(1, 64, seed=11, device=dev)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    ctx, sty = e0.ctx2d[None], e0.style2d[None]
    t_start = time.perf_counter()
    with pytest.warns(RuntimeWarning, match="timed out"):
        toks, n = eng.run(tokens.clone(), batch["masked_perf"], ctx, sty)
    torch.cuda.synchronize()
    assert time.perf_counter() - t_start < 30.0
    assert eng.pair_fallbacks == 1 and eng.pair_groups == 0 and not eng.use_pair
    assert n == n0 and torch.equal(toks, t0)
    for a, b in zip(e0.hid + e0.kc + e0.vc, eng.hid + eng.kc + eng.vc):
        assert torch.equal(a[:n], b[:n])
    toks2, _ = eng.run(tokens.clone(), batch["masked_perf"], ctx, sty)      # and the engine stays usable
    assert torch.equal(toks2, t0) and eng.pair_fallbacks == 1


def test_a_timed_out_hand_off_inside_a_render_session_re_runs_the_notes(dev, monkeypatch):
    """The same fault in the middle of a render session (second window): rows below the window's first new position came from checked
    calls and stay; the faulty steps are re-run through the five launches; every later window uses them too.  Tokens equal a session that
    never had the persistent launch."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import RenderSession
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch
    L = 120
    torch.manual_seed(5)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=13, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    truth, masked = batch["perf"][0], batch["masked_perf"][0]
    ctx, sty = enc.score_embeddings[0], enc.perf_embeddings[0]
    dims = list(PREDICTED_DIMS)
    got = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR", flag)
        sess = RenderSession(model.perf_decoder.model, L, dims)
        rows = []
        for i, k in enumerate((20, 36, 28, 60)):
            g = 8
            win = truth[:k + g].clone()
            win[k:k + g, dims] = 1
            sess.truncate(min(sess.length, k - 1))
            if flag == "1" and i == 1:
                sess.pair_err.fill_(5)                                  # as if a hand-off of this window's first launch had timed out
                with pytest.warns(RuntimeWarning, match="timed out"):
                    rows.append(sess.decode(win, masked[:k + g], ctx[:k + g], sty[:k + g], g).clone())
                assert sess.pair_fallbacks == 1 and sess.pair_groups == 0
            else:
                rows.append(sess.decode(win, masked[:k + g], ctx[:k + g], sty[:k + g], g).clone())
        torch.cuda.synchronize()
        got.append(torch.cat(rows))
    assert torch.equal(got[0], got[1])


def test_head_phase_equals_the_head_launch(dev, monkeypatch):
    """The arg-max LM head as the last phase of the persistent launch (spn_dec_chain_ext.hn > 0) against the same engine with the head in
    its own launch (spn_dec_head): same tokens, same caches; a second run of the same engine (new token buffer, new tables: the head
    record is refreshed per run) gives the same tokens again."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import GreedyDecoder
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    L = 160
    torch.manual_seed(5)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=13, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    monkeypatch.setenv("SPN_DEC_PAIR", "1")
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR_HEAD", flag)
        eng = GreedyDecoder(model.perf_decoder.model, L)
        toks, n = eng.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
        torch.cuda.synchronize()
        assert eng.pair_tail and eng.pair_head == (flag == "1") and int(eng.pair_err.item()) == 0
        res.append((eng, toks.clone(), n))
    (e0, t0, n0), (e1, t1, n1) = res
    assert n0 == n1 == L - 1 and torch.equal(t0, t1)
    assert not (t1[0, 1:, list(e1.cur_dims)] == 1).any()           # every MASK cell of the decoded keys was filled
    for a, b in zip(e0.hid + e0.kc + e0.vc, e1.hid + e1.kc + e1.vc):
        assert torch.equal(a[:n1], b[:n1])
    toks2, n2 = e1.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)   # same engine, fresh buffers
    torch.cuda.synchronize()
    assert n2 == n1 and torch.equal(toks2, t1) and int(e1.pair_err.item()) == 0


@pytest.mark.parametrize("kw", [{}, {"style_emb_mode": "cat"}])
def test_embed_phase_equals_the_embed_launch(dev, monkeypatch, kw):
    """The token-tuple embeddings + their projection as the first phase of the persistent launch, with the NEXT note's AdaLN rows computed
    one note early into the row set of its parity (spn_dec_chain_ext.en / rW), against the same engine with spn_dec_step_begin in front:
    same tokens, same caches, also on a second run of the same engine and from an odd first position."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import GreedyDecoder
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    L = 131
    torch.manual_seed(7)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L, **kw))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=17, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    monkeypatch.setenv("SPN_DEC_PAIR", "1")
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR_EMBED", flag)
        eng = GreedyDecoder(model.perf_decoder.model, L)
        toks, n = eng.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
        torch.cuda.synchronize()
        assert eng.pair_head and eng.pair_embed == (flag == "1") and int(eng.pair_err.item()) == 0
        res.append((eng, toks.clone(), n))
    (e0, t0, n0), (e1, t1, n1) = res
    assert n0 == n1 == L - 1 and torch.equal(t0, t1)
    for a, b in zip(e0.hid + e0.kc + e0.vc + [e0.tok_emb], e1.hid + e1.kc + e1.vc + [e1.tok_emb]):
        assert torch.equal(a[:n1], b[:n1])
    toks2, n2 = e1.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
    torch.cuda.synchronize()
    assert n2 == n1 and torch.equal(toks2, t1) and int(e1.pair_err.item()) == 0


def test_one_launch_note_with_two_embedding_rows_per_wave_and_with_more_keys_than_head_workgroups(dev, monkeypatch):
    """Corners of the phases around the chain: (a) 12 key splits -> 24 attention workgroups for 256 embedding rows: two rows per wave
    (eR = 2); (b) d = 64, one head, all 12 keys decoded -> 8 feed-forward workgroups for 12 keys: a workgroup takes several (key, slab)
    items of the head phase.  Tokens and caches equal the five-launch engine's."""
    (e0, t0, n0), (e1, t1, n1) = _engines(dev, "tiny", 90, monkeypatch, attn_splits=12)
    assert e1.pair_embed and e1.pair_head and e1.pair_chains[0].ext.eR == 2 and int(e1.pair_err.item()) == 0
    assert n0 == n1 and torch.equal(t0, t1)
    for a, b in zip(e0.hid + e0.kc + e0.vc, e1.hid + e1.kc + e1.vc):
        assert torch.equal(a[:n1], b[:n1])
    (e0, t0, n0), (e1, t1, n1) = _engines(dev, "tiny", 70, monkeypatch, mask_all=True, dim=64, heads=1, emb_dims=16)
    assert e1.pair_embed and e1.pair_head and e1.pair_chains[0].ext.hn == 12 and int(e1.pair_err.item()) == 0
    assert e1.g.numel() // 32 < 12                                  # fewer feed-forward workgroups than keys
    assert n0 == n1 and torch.equal(t0, t1) and not (t1[0, 1:] == 1).any()
    for a, b in zip(e0.hid + e0.kc + e0.vc, e1.hid + e1.kc + e1.vc):
        assert torch.equal(a[:n1], b[:n1])


@pytest.mark.parametrize("kw,U", [({}, 16), ({"style_emb_mode": "cat"}, 5), ({}, 3)])
def test_several_notes_per_launch_equal_one_note_per_launch(dev, monkeypatch, kw, U):
    """spn_dec_pairs_notes: U consecutive notes inside ONE launch (the roles loop over the notes; the chosen tokens reach the next note's embed
    phase as granules, key / value rows and the AdaLN rows of the next note cross notes at agent scope) against the same engine with one launch
    per note: same tokens, hidden rows, key / value rows and token embeddings, also on a second run of the same engine."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import GreedyDecoder
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    L = 150
    torch.manual_seed(11)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L, **kw))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=23, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    monkeypatch.setenv("SPN_DEC_PAIR", "1")
    monkeypatch.setenv("SPN_DEC_GRAPH_NOTES", str(U))
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_MULTI_NOTE", flag)
        eng = GreedyDecoder(model.perf_decoder.model, L)
        toks, n = eng.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
        torch.cuda.synchronize()
        assert eng.pair_head and eng.pair_embed and int(eng.pair_err.item()) == 0
        assert eng.pair_chains[0].max_notes == (U if flag == "1" else 1)
        res.append((eng, toks.clone(), n))
    (e0, t0, n0), (e1, t1, n1) = res
    assert n0 == n1 == L - 1 and torch.equal(t0, t1)
    for a, b in zip(e0.hid + e0.kc + e0.vc + [e0.tok_emb], e1.hid + e1.kc + e1.vc + [e1.tok_emb]):
        assert torch.equal(a[:n1], b[:n1])
    toks2, n2 = e1.run(tokens.clone(), batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
    torch.cuda.synchronize()
    assert n2 == n1 and torch.equal(toks2, t1) and int(e1.pair_err.item()) == 0


@pytest.mark.parametrize("temperature,k", [(1.0, None), (0.7, 3)])
def test_sampling_inside_the_launch_draws_what_the_sampling_head_launch_draws(dev, monkeypatch, temperature, k):
    """Top-k sampling as the last phase of the persistent launch (spn_dec_chain_ext.stopk: the slabs hand every logit of their rows to the
    key's first workgroup, which ranks, filters, normalises and draws with the expressions and the counter hash of spn_dec_head_sample)
    against the same session with the sampling head in its own launch: the same tokens draw for draw, over windows that go back and
    re-decode positions, with chord groups long enough for several notes per launch."""
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.decode import RenderSession
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch
    L = 120
    torch.manual_seed(5)
    model = ScorePerformer.init(model_config("tiny", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=13, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    truth, masked = batch["perf"][0], batch["masked_perf"][0]
    ctx, sty = enc.score_embeddings[0], enc.perf_embeddings[0]
    dims = list(PREDICTED_DIMS)
    monkeypatch.setenv("SPN_DEC_PAIR", "1")
    got = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SPN_DEC_PAIR_SAMPLE", flag)
        sess = RenderSession(model.perf_decoder.model, L, dims)
        sess.configure(dict(k=k, thres=0.9, temperature=temperature, seed=41))
        rows = []
        for kk, g in ((20, 8), (36, 8), (28, 20), (60, 3), (44, 8), (90, 24)):
            win = truth[:kk + g].clone()
            win[kk:kk + g, dims] = 1
            sess.truncate(min(sess.length, kk - 1))
            rows.append(sess.decode(win, masked[:kk + g], ctx[:kk + g], sty[:kk + g], g).clone())
        torch.cuda.synchronize()
        assert sess.pair_groups > 0 and sess.pair_head == (flag == "1") and int(sess.pair_err.item()) == 0
        if flag == "1":
            assert sess.pair_embed and sess.pair_chains[0].max_notes > 1       # whole notes, several per launch
        got.append(torch.cat(rows))
    assert torch.equal(got[0], got[1])
    assert not (got[1][:, dims] == 1).any()                                     # every MASK cell was drawn
