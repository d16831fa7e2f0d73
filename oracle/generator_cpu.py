"""CPU restatement of the reference's render loop -- TEST INFRASTRUCTURE, not product code.

Parity PINNED: `tests/golden/render_loop.npz` holds, for three scenarios, every call of the reference's own
`ScorePerformerGenerator.generate_performance_notes` (real reference decoder on CPU fp32, stand-ins of `oracle/render_fakes.py` for
tokenizer / messenger / dataset; `oracle/refimport/make_golden_render.py`); `tests/test_render_cpu.py` replays them through this file.

Cache-free on purpose: the reference's caches (generators.py:215-219,282-287,426-443) change the cost of a call, not its result, so
the oracle decodes every chord group from the whole window with `ref_cpu.greedy_unmask`.  Follows inference/generators.py:
  :128-140  window start at a bar boundary once the accepted sequence reaches max_context_len - 1
  :160-169  chord grouping (equal Bar and Position)
  :177-180  EOS ends the piece
  :184-201  cropping whole bars when the window reaches max_context_len; give up below max_context_len / 8 known notes
  :203-209  bars re-based to the window, masked copy of the window
  :211-213,275-278  delta embedding added to the style rows of the new notes, kept only for accepted notes
  :243-262  onset times from the messenger; stop past start + window + overflow
  :267-280  accept the notes with time <= start + window
"""
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import ref_cpu
from .render_fakes import EOS, MASK, SOS, ZERO


class RenderState:
    def __init__(self, notes: np.ndarray, score_emb: torch.Tensor, perf_emb: torch.Tensor):
        self.notes, self.context, self.embeddings = notes, score_emb, perf_emb.clone()
        self.gen = notes[:1].copy()
        self.reached_eos = False


def render_window(sd, cfg, st: RenderState, times_fn: Callable, mask_dims: List[int], *, start_time: float, time_window: float,
                  time_window_overflow: float, delta: Optional[torch.Tensor], max_context_len: int, group_chord_notes: bool
                  ) -> Tuple[Optional[np.ndarray], int]:
    """One `generate_performance_notes` call.  Returns (accepted tokens or None, number of decoded notes)."""
    notes, acc = st.notes, st.gen
    cur = len(acc)
    start = 0
    if cur >= max_context_len - 1:
        nb = np.nonzero(np.diff(acc[1:, 0]))[0]
        if len(nb):
            fits = np.nonzero(cur - (nb + 1) < max_context_len)[0]
            start = 0 if len(fits) == 0 else int(nb[fits[0]]) + 2
    win = acc[start:].copy()
    known, first = len(win), int(acc[start, 0] == SOS)
    style = st.embeddings.clone()
    times, toks = [], []
    while not st.reached_eos:
        end = cur + 1
        while group_chord_notes and end < len(notes) and (notes[cur, :2] == notes[end, :2]).all():
            end += 1
        new = notes[cur:end]
        if new[-1, 0] == EOS:
            st.reached_eos = True
            break
        win = np.concatenate([win, new])
        last = len(win)
        if last >= max_context_len:
            nb = np.nonzero(np.diff(win[first:last, 0]))[0]
            shift = 1
            if len(nb):
                fits = np.nonzero(last - (nb + first) < max_context_len)[0]
                if len(fits) and nb[fits[0]] + 1 + first != last - 1:
                    shift = int(nb[fits[0]]) + 1 + first
            win, known, start, first = win[shift:], known - shift, start + shift, 0
            last = len(win)
            if known < max_context_len / 8:
                break
        base = win[first, 0] - ZERO
        x = win.copy()
        x[first:last, 0] -= base
        xm = x.copy()
        xm[first:last, mask_dims] = MASK
        if delta is not None:
            style[cur:end] += delta
        out = ref_cpu.greedy_unmask(sd, cfg, torch.from_numpy(x)[None], torch.from_numpy(xm)[None], st.context[start:end][None],
                                    style[start:end][None])[0].numpy()
        g = out[-len(new):].copy()
        g[:, 0] += base
        t = times_fn(g)
        times.extend(t.tolist())
        toks.append(g)
        if t.max() >= start_time + time_window + time_window_overflow:
            break
        win[-len(new):] = g
        cur = end
    if not toks:
        return None, 0
    keep = np.nonzero(np.asarray(times) <= start_time + time_window)[0]
    cut = 0 if len(keep) == 0 else int(keep[-1]) + 1
    if cut == 0:
        return None, len(times)
    accepted = np.concatenate(toks)[:cut]
    total = len(acc)
    if delta is not None:
        st.embeddings[total:total + cut] = style[total:total + cut]
    st.gen = np.concatenate([acc, accepted])
    return accepted, len(times)
