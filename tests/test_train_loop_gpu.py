"""GPU: the pieces compose like the reference's training loop (trainer.py:440-470): ragged samples -> device-side collator ->
prepare_inputs -> model -> attached evaluator -> clip + AdamW, with a checkpoint written and resumed in the middle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = dict(pad_token_id=0, pad_to_multiple_of=8, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
          mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9])
IGNORE = ["Bar", "Position", "Pitch", "Duration", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset"]


def make_samples(rng, sizes, lengths):
    out = []
    for n in lengths:
        perf = np.stack([rng.integers(4, v, size=n) for v in sizes], -1).astype(np.int64)
        perf[0] = 2
        seg = SimpleNamespace(**{k: (np.arange(n) // d + 4).astype(np.int64) for k, d in (("bar", 8), ("beat", 4), ("onset", 2))})
        out.append(SimpleNamespace(score=perf[:, :10].copy(), perf=perf, noisy_perf=None, directions=None, segments=seg, is_deadpan=False))
    return out


def test_collate_train_evaluate_checkpoint_resume(tmp_path):
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.checkpoint import load_checkpoint, save_checkpoint
    from scoreperformer_amd.data import MixedLMScorePerformanceCollator
    from scoreperformer_amd.models import ScorePerformer, ScorePerformerEvaluator
    from scoreperformer_amd.synthetic import model_config, PERFORMANCE_VOCAB
    dev = torch.device("cuda")
    cfg = model_config("tiny", dropout=0.0)
    sizes = list(PERFORMANCE_VOCAB.values())
    rng = np.random.default_rng(0)
    batches = [make_samples(rng, sizes, lens) for lens in ([40, 64, 33], [57, 20, 48], [64, 64, 9], [31, 45, 60])]
    collate = MixedLMScorePerformanceCollator(**KW)

    def make():
        model = ScorePerformer.init(model_config("tiny", dropout=0.0))
        model.load_state_dict(filled_state_dict(model, seed=4))
        arena = ParamArena(model, dev)
        model.train()
        opt = FusedAdamW(arena, lr=1e-3, weight_decay=1e-2, grad_clip=2.0)
        ev = ScorePerformerEvaluator(model, ignore_keys=IGNORE, weighted_distance=True,
                                     token_values={k: torch.linspace(0, 1, v).tolist() for k, v in cfg["num_tokens"].items()}).attach()
        return model, arena, opt, ev

    def step(model, arena, opt, ev, samples, seed):
        inputs = model.allocate_inputs(model.prepare_inputs(collate(samples)), dev)
        torch.manual_seed(seed)                                    # the MMD prior draws
        out = model(**inputs)
        metrics = ev(inputs, out)
        arena.zero_grad()
        out.loss.backward()
        opt.step()
        assert torch.isfinite(out.loss) and all(torch.isfinite(v) for v in metrics.values())
        assert {"accuracy", "accuracy/pred", "accuracy/Velocity", "distance/Tempo"} <= set(metrics)
        return float(out.loss.detach()), {k: float(v) for k, v in metrics.items()}

    # uninterrupted run
    m1 = make()
    ref = [step(*m1, b, 10 + i) for i, b in enumerate(batches)]
    # same run, checkpointed after two steps and resumed in a fresh model / arena / optimizer
    m2 = make()
    got = [step(*m2, b, 10 + i) for i, b in enumerate(batches[:2])]
    path = str(tmp_path / "mid.pt")
    save_checkpoint(path, m2[0], m2[2], model_config=cfg)
    m3 = make()
    load_checkpoint(path, m3[0], m3[2])
    got += [step(*m3, b, 12 + i) for i, b in enumerate(batches[2:])]
    for (la, ma), (lb, mb) in zip(ref, got):
        assert abs(la - lb) < 2e-3 * abs(la), (la, lb)             # float atomics (segment sums, embedding scatter): order noise, amplified over steps
        assert sorted(ma) == sorted(mb)
        for k in ma:
            assert abs(ma[k] - mb[k]) < 1e-2 * max(1.0, abs(ma[k])), k
    assert ref[-1][0] < ref[0][0] + 1.0                            # and it trains (loss does not blow up over the four steps)


def test_composed_train_step_makes_no_host_synchronisation():
    """The collator hands the segment slot counts over as python ints (`segments.bounds`, data.SegmentBounds; the reference reads
    `segments.max() + 1` back from the device in every forward, mmd_transformer.py:330) and `model.sync_free` keeps every loss key, so
    forward + backward + clip + AdamW of a collated batch enqueue work and never wait for the device: run under
    torch.cuda.set_sync_debug_mode("error"), where any synchronising call raises."""
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.data import MixedLMScorePerformanceCollator, SegmentBounds
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, PERFORMANCE_VOCAB
    dev = torch.device("cuda")
    rng = np.random.default_rng(1)
    sizes = list(PERFORMANCE_VOCAB.values())
    collate = MixedLMScorePerformanceCollator(**KW)
    model = ScorePerformer.init(model_config("tiny", dropout=0.1))
    model.load_state_dict(filled_state_dict(model, seed=4))
    arena = ParamArena(model, dev)
    model.train()
    model.sync_free = True
    opt = FusedAdamW(arena, lr=1e-3, weight_decay=1e-2, grad_clip=2.0)
    losses = []
    for i, lens in enumerate(([40, 64, 33], [57, 20, 48], [64, 64, 9])):
        samples = make_samples(rng, sizes, lens)
        collated = collate(samples)
        bounds = collated.segments.bounds
        assert isinstance(bounds, SegmentBounds)
        assert bounds == {k: int(max(getattr(s.segments, k).max() for s in samples)) + 1 for k in ("bar", "beat", "onset")}
        inputs = model.allocate_inputs(model.prepare_inputs(collated), dev)      # the trainer's two calls (trainer.py:440-447)
        assert inputs["segment_bounds"] is bounds
        torch.manual_seed(20 + i)
        if i == 0:      # first step un-guarded: lazy one-time set-up (library load, workspaces, constant tables) may synchronise
            out = model(**inputs); out.loss.backward(); opt.step()
            torch.cuda.synchronize()
            continue
        torch.cuda.set_sync_debug_mode("error")
        try:
            out = model(**inputs)
            out.loss.backward()
            opt.step()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        losses.append(float(out.loss.detach()))
    assert len(losses) == 2 and all(np.isfinite(v) for v in losses)
    # the bounds only size the segment slots: the same batch without them (one host read instead) gives the same loss
    model.eval()
    with torch.no_grad():
        torch.manual_seed(22)
        a = float(model(**inputs).loss)
        torch.manual_seed(22)
        b = float(model(**{k: v for k, v in inputs.items() if k != "segment_bounds"}).loss)
    assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)
