"""CPU: the render-loop oracle (oracle/generator_cpu.py) replays the reference generator's golden calls token for token."""
import numpy as np
import pytest
import torch

from render_common import MASK_DIMS, load

VOCAB, SEED, SCEN = load()


@pytest.mark.parametrize("name", sorted(SCEN))
def test_oracle_render_loop_matches_the_reference_generator(name):
    from oracle.generator_cpu import RenderState, render_window
    from oracle.render_fakes import FakeMessenger
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    s = SCEN[name]
    cfg = model_config(preset="tiny", num_tokens=VOCAB)
    sd = filled_state_dict(ScorePerformer.init(model_config(preset="tiny", num_tokens=VOCAB)), seed=SEED)
    st = RenderState(s["notes"].copy(), torch.from_numpy(s["score_emb"]), torch.from_numpy(s["perf_emb"]))
    times = FakeMessenger(VOCAB).times
    c = s["cfg"]
    t = 0.0
    for i, call in enumerate(s["calls"]):
        delta = torch.from_numpy(s["delta"]) if c["delta_every"] and i % c["delta_every"] == 0 else None
        got, _ = render_window(sd, cfg, st, times, MASK_DIMS, start_time=t, time_window=c["time_window"],
                               time_window_overflow=c["time_window_overflow"], delta=delta, max_context_len=c["max_context_len"],
                               group_chord_notes=c["group_chord_notes"])
        want = call["tokens"]
        assert (got is None and len(want) == 0) or np.array_equal(got, want), (name, i)
        t += c["time_window"]
    assert st.reached_eos and np.array_equal(st.gen, s["gen_seq"])
    assert np.allclose(st.embeddings.numpy(), s["final_embeddings"], atol=1e-6)
