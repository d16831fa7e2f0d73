"""Deterministic stand-ins for the objects the render loop talks to (tokenizer, messenger, dataset) -- TEST INFRASTRUCTURE.

The reference's `ScorePerformerGenerator` (inference/generators.py:35-60) takes a dataset, a collator and a MIDI messenger built on
miditok / mido, which are not available here.  The loop itself only needs: token ids of SOS/EOS and the zero token, a function from
generated tokens to onset times (the messenger), and the token arrays of a piece (the dataset).  These minimal, build-owned fakes
provide exactly that surface to BOTH the real reference (in `oracle/refimport/make_golden_render.py`) and this repository's loop, so
the golden vectors pin the loop logic: chord grouping, context cropping at bar boundaries, bar re-basing, cache reuse/cutting, delta
embeddings, time-window cutting.  No reference code here.
"""
from types import SimpleNamespace

import numpy as np

KEYS = ["Bar", "Position", "Pitch", "Velocity", "Duration", "Tempo", "TimeSig", "PositionShift", "NotesInOnset", "PositionInOnset",
        "RelOnsetDev", "RelPerfDuration"]
PAD, MASK, SOS, EOS, ZERO = 0, 1, 2, 3, 4
BAR_TICKS, TICK = 32, 0.02


class FakeTokenizer:
    zero_token = ZERO

    def __init__(self, vocab):
        self.sizes = [vocab[k] for k in KEYS]
        self.vocab_types_idx = {k: i for i, k in enumerate(KEYS)}

    def __getitem__(self, item):                      # tokenizer[0, "SOS_None"] / tokenizer[0, "EOS_None"]
        _, name = item
        return {"SOS_None": SOS, "EOS_None": EOS, "PAD_None": PAD, "MASK_None": MASK}[name]


class FakeMessenger:
    """Onset time of a note = its bar/position on a fixed grid + a deviation read from the PREDICTED RelOnsetDev token, so the
    loop's control flow depends on what the decoder generated."""

    def __init__(self, vocab):
        self.center = (vocab["RelOnsetDev"] - ZERO) // 2

    def times(self, tokens):
        tokens = np.asarray(tokens)
        grid = ((tokens[:, 0] - ZERO) * BAR_TICKS + (tokens[:, 1] - ZERO)) * TICK
        return np.maximum(grid + (tokens[:, 10] - ZERO - self.center) * 0.004, 0.0)

    def tokens_to_messages(self, tokens, note_attributes=True, note_off_events=True, intermediates=None, return_intermediates=False,
                           to_times=True, sort=True):
        t = self.times(tokens)
        if not note_attributes:
            out = t
        else:
            out = [(round(float(ti), 6), int(tok[2]), int(tok[3]), int(tok[11])) for ti, tok in zip(t, np.asarray(tokens))]
            if sort:
                out = sorted(out)
        return (out, intermediates) if return_intermediates else out


class FakeProcessor:
    @staticmethod
    def add_sos_token(seq):
        return np.concatenate([np.full((1, seq.shape[1]), SOS, seq.dtype), seq])

    @staticmethod
    def add_eos_token(seq):
        return np.concatenate([seq, np.full((1, seq.shape[1]), EOS, seq.dtype)])


def make_piece(seed, n_notes, vocab, notes_per_bar=6.0):
    """A performance token array [n, 12]: bars/positions non-decreasing, chords (equal bar+position) of 1-3 notes."""
    rng = np.random.default_rng(seed)
    seq = np.stack([rng.integers(ZERO, vocab[k], size=n_notes) for k in KEYS], -1).astype(np.int64)
    bar, pos, i = 0, 0, 0
    while i < n_notes:
        chord = int(rng.integers(1, 4))
        seq[i:i + chord, 0], seq[i:i + chord, 1] = ZERO + bar, ZERO + pos
        i += chord
        pos += int(rng.integers(2, 2 * int(BAR_TICKS / notes_per_bar) + 1))
        if pos >= BAR_TICKS:
            bar, pos = bar + 1, pos - BAR_TICKS
    assert seq[:, 0].max() < vocab["Bar"] and seq[:, 1].max() < vocab["Position"]
    return seq


def make_dataset(vocab, pieces):
    return SimpleNamespace(tokenizer=FakeTokenizer(vocab), performances=list(pieces), processor=FakeProcessor(),
                           performance_names=[f"piece{i}" for i in range(len(pieces))])


class _Scores(list):
    def __init__(self, seqs, name_to_idx):
        super().__init__(seqs)
        self._name_to_idx = name_to_idx


# ---- a windowed score/performance dataset for `encode_embeddings` (generators.py:320-424) ----------------------------------------
class FakeScoreDataset:
    """One aligned score/performance pair served in bar windows, with the attributes `encode_embeddings` reads from the reference's
    `ScorePerformanceDataset`: performance_names, _performance_map, scores (+ _name_to_idx), _score_indices, indexer, get(meta),
    max_seq_len, max_bar, _beat_maps, _onset_maps, tokenizer, performances, processor."""

    def __init__(self, vocab, perf_seq, max_seq_len=48, max_bar=256):
        self.tokenizer = FakeTokenizer(vocab)
        self.processor = FakeProcessor()
        self.performances = [perf_seq]
        self.performance_names = ["perf0"]
        self._performance_map = {"perf0": ("score0", None)}
        score_seq = perf_seq[:, :10].copy()
        self.scores = _Scores([score_seq], {"score0": 0})
        self._score_indices = [None]
        self.indexer = SimpleNamespace(compute_bar_indices=self._bar_indices)
        self.max_seq_len, self.max_bar = max_seq_len, max_bar
        n = len(perf_seq)
        self._beat_maps = [np.cumsum(np.r_[0, np.diff(perf_seq[:, 1]) != 0]).astype(np.int64) // 2 + ZERO]
        self._onset_maps = [np.cumsum(np.r_[0, (np.diff(perf_seq[:, 0]) != 0) | (np.diff(perf_seq[:, 1]) != 0)]).astype(np.int64) + ZERO]
        assert len(self._beat_maps[0]) == n

    @staticmethod
    def _bar_indices(score_seq):
        bars = score_seq[:, 0] - ZERO
        first = [int(np.argmax(bars >= b)) for b in range(int(bars.max()) + 1)]
        return np.array(first + [len(score_seq)], dtype=np.int64)       # first note of every bar, then the total

    def get(self, meta):
        perf, score = self.performances[0], self.scores[0]
        idx = self._bar_indices(score)
        total_bars = len(idx) - 2
        lo, hi = idx[meta.start_bar], idx[min(meta.end_bar, total_bars) + 1]
        s, p = score[lo:hi].copy(), perf[lo:hi].copy()
        seg = SimpleNamespace(bar=(s[:, 0] - s[0, 0] + ZERO).astype(np.int64), beat=(self._beat_maps[0][lo:hi] - self._beat_maps[0][lo] + ZERO),
                              onset=(self._onset_maps[0][lo:hi] - self._onset_maps[0][lo] + ZERO))
        pad = lambda a, left, right: np.concatenate([[a[0]]] * left + [a] + [[a[-1]]] * right)
        sos, eos = meta.start_bar == 0, meta.end_bar >= total_bars
        if sos:
            s, p = FakeProcessor.add_sos_token(s), FakeProcessor.add_sos_token(p)
        if eos:
            s, p = FakeProcessor.add_eos_token(s), FakeProcessor.add_eos_token(p)
        seg = SimpleNamespace(**{k: pad(getattr(seg, k), int(sos), int(eos)) for k in ("bar", "beat", "onset")})
        return SimpleNamespace(score=s, perf=p, noisy_perf=None, segments=seg, directions=None, is_deadpan=False, meta=meta)
