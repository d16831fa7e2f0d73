"""Golden vectors of the reference's MixedLMScorePerformanceCollator (run in the authoring container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.refimport.make_golden_collate

Writes tests/golden/collate_mixlm.npz: ragged random samples (inputs) and the tensors the REAL reference collator returns
for several constructor settings.  Data only; no reference source is stored.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden")   # module-level, like the other generators: a checker may point it elsewhere
sys.path.insert(0, ROOT)
from oracle.refimport import stubs  # noqa: E402

stubs.install()
sys.path.insert(0, stubs.REFERENCE_ROOT)
from scoreperformer.data.collators.score_performance import MixedLMScorePerformanceCollator  # noqa: E402

CASES = {
    # name: (collator kwargs, lengths, inference)
    "recipe": (dict(pad_token_id=0, pad_to_multiple_of=1, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
                    mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9]), [37, 64, 5, 50], False),      # recipes/scoreperformer/base.yaml:60-66
    "pad8_labels_all_dims": (dict(pad_token_id=0, pad_to_multiple_of=8, mask_token_id=1, mask_ignore_token_ids=[2, 3],
                                  mask_ignore_token_dims=[0, 2], label_pad_ignored_dims=False), [13, 9, 30], False),
    "inference_no_dims": (dict(pad_token_id=0, pad_to_multiple_of=16, mask_token_id=1), [21, 1, 17, 2, 33], True),
    "noisy_pad4": (dict(pad_token_id=0, pad_to_multiple_of=4, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
                        mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9]), [19, 7, 26], False),   # samples carry a noisy performance
    "single": (dict(pad_token_id=0, pad_to_multiple_of=1, mask_token_id=1, mask_ignore_token_ids=[0, 1, 2, 3],
                    mask_ignore_token_dims=[0, 1, 2, 4, 6, 7, 8, 9], label_pad_token_id=-7), [11], False),
}


def make_samples(lengths, rng, ks=10, kp=12, noisy=False):
    samples = []
    for n in lengths:
        score = rng.integers(0, 40, size=(n, ks)).astype(np.int64)
        perf = rng.integers(0, 40, size=(n, kp)).astype(np.int64)
        score[0], perf[0] = 2, 2                       # SOS rows like the dataset builds them
        if n > 2:
            score[-1], perf[-1] = 3, 3                 # EOS rows
        bar = np.cumsum(rng.integers(0, 2, size=n)).astype(np.int64) + 1
        beat = np.cumsum(rng.integers(0, 2, size=n)).astype(np.int64) + 1
        onset = np.cumsum(rng.integers(0, 2, size=n)).astype(np.int64) + 1
        noisy_perf = rng.integers(0, 40, size=(int(rng.integers(1, n + 6)), kp)).astype(np.int64) if noisy else None
        samples.append(SimpleNamespace(score=score, perf=perf, noisy_perf=noisy_perf, directions=None, is_deadpan=bool(rng.integers(0, 2)),
                                       segments=SimpleNamespace(bar=bar, beat=beat, onset=onset)))
    return samples


def main():
    rng = np.random.default_rng(20240917)
    out = {}
    for name, (kw, lengths, inference) in CASES.items():
        samples = make_samples(lengths, rng, noisy=name.startswith("noisy"))
        data = MixedLMScorePerformanceCollator(**kw)(samples, inference=inference)
        out[f"{name}/kwargs"] = np.array(repr(dict(kw, inference=inference)))
        for i, smp in enumerate(samples):
            out[f"{name}/in/score{i}"], out[f"{name}/in/perf{i}"] = smp.score, smp.perf
            out[f"{name}/in/bar{i}"], out[f"{name}/in/beat{i}"], out[f"{name}/in/onset{i}"] = smp.segments.bar, smp.segments.beat, smp.segments.onset
        out[f"{name}/in/deadpan"] = np.array([s.is_deadpan for s in samples])
        if samples[0].noisy_perf is not None:
            for i, smp in enumerate(samples):
                out[f"{name}/in/noisy{i}"] = smp.noisy_perf
        ref = {"score": data.scores.tokens, "score_mask": data.scores.mask, "score_len": data.scores.lengths,
               "perf": data.performances.tokens, "perf_mask": data.performances.mask, "perf_len": data.performances.lengths,
               "masked_perf": data.masked_performances.tokens, "labels": data.labels.tokens, "labels_mask": data.labels.mask,
               "bar": data.segments.bar, "beat": data.segments.beat, "onset": data.segments.onset, "deadpan_mask": data.deadpan_mask}
        if data.noisy_performances is not None:
            ref.update(noisy=data.noisy_performances.tokens, noisy_mask=data.noisy_performances.mask, noisy_len=data.noisy_performances.lengths)
        for k, v in ref.items():
            out[f"{name}/out/{k}"] = v.numpy()
    path = os.path.join(OUT, "collate_mixlm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
