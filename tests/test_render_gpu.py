"""GPU: the product render loop (scoreperformer_amd.inference.ScorePerformerGenerator over decode.RenderSession) replays the
reference generator's golden calls: tokens bit-exact, same messages, same accepted-note bookkeeping."""
import numpy as np
import pytest
import torch

from render_common import COLLATOR, load

pytestmark = pytest.mark.gpu
VOCAB, SEED, SCEN = load()


def make(use_engine, dev):
    from types import SimpleNamespace
    from oracle.render_fakes import FakeMessenger, make_dataset
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.inference import ScorePerformerGenerator
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=VOCAB))
    model.load_state_dict(filled_state_dict(model, seed=SEED))
    arena = ParamArena(model, dev)
    model.eval()
    collator = SimpleNamespace(mask_token_id=COLLATOR["mask_token_id"], mask_ignore_token_dims=COLLATOR["mask_ignore_token_dims"])

    def gen_for(piece):
        return ScorePerformerGenerator(model, make_dataset(VOCAB, [piece]), collator, FakeMessenger(VOCAB), device=dev, use_engine=use_engine)
    return gen_for, arena


def replay(gen, s, dev):
    from scoreperformer_amd.modules.sampling import top_k
    c = s["cfg"]
    gen.prepare_performance_notes(0, score_embeddings=torch.from_numpy(s["score_emb"]).clone(), perf_embeddings=torch.from_numpy(s["perf_emb"]).clone())
    assert np.array_equal(gen.perf_data.notes.cpu().numpy(), s["notes"])
    t = 0.0
    for i, call in enumerate(s["calls"]):
        delta = torch.from_numpy(s["delta"]).clone() if c["delta_every"] and i % c["delta_every"] == 0 else None
        seq, messages = gen.generate_performance_notes(
            start_time=t, time_window=c["time_window"], time_window_overflow=c["time_window_overflow"], delta_embedding=delta,
            max_context_len=c["max_context_len"], group_chord_notes=c["group_chord_notes"], filter_logits_fn=top_k, filter_kwargs={"k": 1})
        want = call["tokens"]
        if len(want) == 0:
            assert seq is None and messages == [], i
        else:
            assert seq is not None and np.array_equal(seq.cpu().numpy(), want), i          # bit-exact greedy tokens
            assert np.allclose(np.array(messages, np.float64).reshape(-1, 4), call["messages"], atol=1e-9), i
        assert int(gen.predict_number_of_notes(start_time=t + c["time_window"], time_window=c["time_window"])) == call["predicted_notes"], i
        t += c["time_window"]
        yield i, call
    assert gen.perf_data.reached_eos
    assert np.array_equal(gen.perf_data.gen_seq.cpu().numpy(), s["gen_seq"])
    assert np.allclose(gen.perf_data.embeddings.cpu().numpy(), s["final_embeddings"], atol=1e-6)


@pytest.mark.parametrize("name", sorted(SCEN))
def test_engine_render_loop_matches_the_reference_generator(name):
    dev = torch.device("cuda")
    gen_for, _ = make(True, dev)
    s = SCEN[name]
    gen = gen_for(s["piece"])
    for i, call in replay(gen, s, dev):
        # the static cache holds exactly the rows the reference's concatenated caches hold after the call
        have = -1 if gen.perf_data.caches is None else gen.perf_data.caches.length
        assert have == call["cache_len"], (i, have, call["cache_len"])
    assert gen._session is not None and gen._session.steps_run > 0, "the HIP decode engine must have produced the tokens"
    if name == "crop":
        assert gen._session.prefilled_rows > 0          # cropped windows were re-primed by the batched fp32 pass


def test_sequential_prefill_gives_the_same_tokens():
    """prefill="sequential" (note-by-note recompute after a crop) and the batched fp32 prefill agree with the reference and each other."""
    dev = torch.device("cuda")
    gen_for, _ = make(True, dev)
    s = SCEN["crop"]
    gen = gen_for(s["piece"])
    gen.prefill = "sequential"
    list(replay(gen, s, dev))
    assert gen._session.prefilled_rows == 0 and gen._session.steps_run > 0


def test_module_path_render_loop_runs_the_same_loop():
    """use_engine=False: `perf_decoder.unmask_tokens` with TupleTransformerCaches + cut_caches, as the reference calls it.  The module
    path computes its GEMMs in bf16, so near-tied arg-maxes differ from the fp32 reference and the differences feed back into the
    context: the score-given dims must be exact, and the loop with caches (reuse, cutting, rebuilding after a crop) must produce
    exactly what the loop without caches produces."""
    from scoreperformer_amd.modules.sampling import top_k
    dev = torch.device("cuda")
    s = SCEN["single_notes"]
    c = s["cfg"]
    outs = []
    for disable_caches in (False, True):
        gen_for, _ = make(False, dev)
        gen = gen_for(s["piece"])
        gen.model.perf_decoder.use_decode_engine = False
        gen.prepare_performance_notes(0, score_embeddings=torch.from_numpy(s["score_emb"]).clone(),
                                      perf_embeddings=torch.from_numpy(s["perf_emb"]).clone())
        t, calls, cached_calls = 0.0, 0, 0
        while not gen.perf_data.reached_eos and calls < 100:
            gen.generate_performance_notes(start_time=t, time_window=c["time_window"], time_window_overflow=c["time_window_overflow"],
                                           max_context_len=c["max_context_len"], group_chord_notes=False, filter_logits_fn=top_k,
                                           filter_kwargs={"k": 1}, disable_caches=disable_caches)
            if gen.perf_data.caches is not None:
                cached_calls += 1
                start = gen._window_start(gen._gen, c["max_context_len"])
                assert gen.perf_data.caches.token_emb.shape[1] <= len(gen._gen) - start - 1
            t += c["time_window"]
            calls += 1
        assert gen._session is None and gen.perf_data.reached_eos and (disable_caches or cached_calls > 0)
        outs.append(gen.perf_data.gen_seq.cpu().numpy())
    want = s["gen_seq"]
    given = [d for d in range(12) if d not in (3, 5, 10, 11)]
    assert outs[0].shape == want.shape and np.array_equal(outs[0][:, given], want[:, given])
    assert np.array_equal(outs[0], outs[1])
    assert (outs[0] == want).mean() > 0.85


def test_engine_window_must_fit():
    dev = torch.device("cuda")
    gen_for, _ = make(True, dev)
    s = SCEN["single_notes"]
    gen = gen_for(s["piece"])
    gen.engine_max_len = 64
    gen.prepare_performance_notes(0, score_embeddings=torch.from_numpy(s["score_emb"]), perf_embeddings=torch.from_numpy(s["perf_emb"]))
    from scoreperformer_amd.modules.sampling import top_k
    with pytest.raises(ValueError):
        gen.generate_performance_notes(max_context_len=512, filter_logits_fn=top_k, filter_kwargs={"k": 1})


def test_batched_prefill_adopts_module_caches():
    """prefill="modules": after a crop the window's caches come from one batched module forward (bf16) and the session continues
    from them.  Same loop, same bookkeeping; tokens may differ from the fp32 reference only through bf16 near-ties."""
    from scoreperformer_amd.modules.sampling import top_k
    dev = torch.device("cuda")
    gen_for, _ = make(True, dev)
    s = SCEN["crop"]
    c = s["cfg"]
    gen = gen_for(s["piece"])
    gen.prefill, gen.prefill_min = "modules", 8
    gen.prepare_performance_notes(0, score_embeddings=torch.from_numpy(s["score_emb"]).clone(), perf_embeddings=torch.from_numpy(s["perf_emb"]).clone())
    t, calls = 0.0, 0
    while not gen.perf_data.reached_eos and calls < 200:
        gen.generate_performance_notes(start_time=t, time_window=c["time_window"], time_window_overflow=c["time_window_overflow"],
                                       max_context_len=c["max_context_len"], filter_logits_fn=top_k, filter_kwargs={"k": 1})
        t += c["time_window"]
        calls += 1
    got, want = gen.perf_data.gen_seq.cpu().numpy(), s["gen_seq"]
    given = [d for d in range(12) if d not in (3, 5, 10, 11)]
    assert gen.perf_data.reached_eos and got.shape == want.shape and np.array_equal(got[:, given], want[:, given])
    # far fewer sequential steps than notes x window: the crops were served by batched forwards
    exact = gen_for(s["piece"])
    exact.prefill = "sequential"
    list(replay(exact, s, dev))
    assert gen._session.steps_run < 0.5 * exact._session.steps_run
    assert (got == want).mean() > 0.85


@pytest.mark.parametrize("overlay", [0.0, 0.5])
def test_encode_embeddings_windows_match_the_reference(overlay):
    """`encode_embeddings` (generators.py:320-424): bar windows of `dataset.max_seq_len` notes through the HIP encoders (batches built
    by the device-side collator), overlapping parts dropped, against the reference's concatenated embeddings and latents."""
    import os
    from oracle.render_fakes import FakeMessenger, FakeScoreDataset
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.data import MixedLMScorePerformanceCollator
    from scoreperformer_amd.inference import ScorePerformerGenerator
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "encode_embeddings.npz"))
    dev = torch.device("cuda")
    model = ScorePerformer.init(model_config(preset="tiny", num_tokens=VOCAB))
    model.load_state_dict(filled_state_dict(model, seed=SEED))
    arena = ParamArena(model, dev)   # noqa: F841
    model.eval()
    ds = FakeScoreDataset(VOCAB, z["piece"], max_seq_len=int(z["max_seq_len"]))
    gen = ScorePerformerGenerator(model, ds, MixedLMScorePerformanceCollator(**COLLATOR), FakeMessenger(VOCAB), device=dev)
    se, pe, lat = gen.encode_embeddings(0, compute_latents=True, overlay_bars=overlay)
    for got, key in ((se, "score_emb"), (pe, "perf_emb")):
        want = z[f"overlay{overlay}/{key}"]
        got = got.float().cpu().numpy()
        assert got.shape == want.shape                                  # the window bookkeeping: every note exactly once
        assert np.abs(got - want).max() <= 0.03 * np.abs(want).max(), key   # bf16 encoder GEMMs
    lat = lat if isinstance(lat, (list, tuple)) else [lat]
    for i, t in enumerate(lat):
        want = z[f"overlay{overlay}/latent{i}"]
        assert tuple(t.shape) == want.shape
        assert np.abs(t.float().cpu().numpy() - want).max() <= 0.05 * max(np.abs(want).max(), 1e-3), i
    # prepare_performance_notes computes them itself when they are not passed (generators.py:87-91)
    gen.prepare_performance_notes(0, overlay_bars=overlay)
    assert gen.perf_data.embeddings.shape[0] == len(z["piece"]) + 2 and gen.perf_data.context.shape[0] == len(z["piece"]) + 2


def test_engine_render_loop_with_top_k_sampling():
    """`filter_logits_fn=top_k` with k > 1 (the reference's default is sampling): the session samples on the device.  Draws cannot
    match torch.multinomial's stream, so: runs to EOS, score-given dims exact, predicted tokens valid ids, same seed -> same piece,
    other seed -> another piece, and it is not the greedy piece."""
    from scoreperformer_amd.modules.sampling import top_k
    dev = torch.device("cuda")
    gen_for, _ = make(True, dev)
    s = SCEN["long_context"]
    c = s["cfg"]
    given = [d for d in range(12) if d not in (3, 5, 10, 11)]
    sizes = list(VOCAB.values())

    def render(seed, kw):
        gen = gen_for(s["piece"])
        gen.sampling_seed = seed
        gen.prepare_performance_notes(0, score_embeddings=torch.from_numpy(s["score_emb"]).clone(), perf_embeddings=torch.from_numpy(s["perf_emb"]).clone())
        t, calls = 0.0, 0
        while not gen.perf_data.reached_eos and calls < 300:
            gen.generate_performance_notes(start_time=t, time_window=c["time_window"], time_window_overflow=c["time_window_overflow"],
                                           max_context_len=c["max_context_len"], filter_logits_fn=top_k, filter_kwargs=kw)
            t += c["time_window"]
            calls += 1
        assert gen.perf_data.reached_eos and gen._session is not None and gen._session.steps_run > 0
        return gen.perf_data.gen_seq.cpu().numpy()

    a, b, other, greedy = render(1, {"k": 4}), render(1, {"k": 4}), render(2, {"k": 4}), render(1, {"k": 1})
    assert np.array_equal(greedy, s["gen_seq"])
    assert a.shape == s["gen_seq"].shape and np.array_equal(a[:, given], s["gen_seq"][:, given])
    for d in (3, 5, 10, 11):
        assert a[1:, d].min() >= 2 and a[:, d].max() < sizes[d]
    assert np.array_equal(a, b) and not np.array_equal(a, other) and not np.array_equal(a, greedy)
    default = render(3, None)                                    # thres 0.9: k = ceil(0.1 * V) per key
    assert default.shape == a.shape
