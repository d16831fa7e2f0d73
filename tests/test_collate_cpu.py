"""CPU: the collator oracle (oracle/collate_cpu.py) against the reference collator's own outputs (tests/golden/collate_mixlm.npz)."""
import ast
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "collate_mixlm.npz")


def load_cases():
    z = np.load(GOLD, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in z.files})
    cases = {}
    for name in names:
        kw = ast.literal_eval(str(z[f"{name}/kwargs"]))
        n = len([k for k in z.files if k.startswith(f"{name}/in/score")])
        ins = dict(scores=[z[f"{name}/in/score{i}"] for i in range(n)], perfs=[z[f"{name}/in/perf{i}"] for i in range(n)],
                   segments=[{s: z[f"{name}/in/{s}{i}"] for s in ("bar", "beat", "onset")} for i in range(n)],
                   deadpan=z[f"{name}/in/deadpan"].tolist(),
                   noisy=[z[f"{name}/in/noisy{i}"] for i in range(n)] if f"{name}/in/noisy0" in z.files else None)
        outs = {k.split("/out/")[1]: z[k] for k in z.files if k.startswith(f"{name}/out/")}
        cases[name] = (kw, ins, outs)
    return cases


CASES = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_collator_is_bit_exact_against_the_reference(name):
    from oracle.collate_cpu import collate_mixlm
    kw, ins, ref = CASES[name]
    got = collate_mixlm(ins["scores"], ins["perfs"], ins["segments"], ins["deadpan"], noisy=ins["noisy"], **kw)
    for key, want in ref.items():
        have = got["perf_mask"] if key == "labels_mask" else got[key]
        assert have.shape == want.shape and have.dtype == want.dtype, (key, have.shape, want.shape, have.dtype, want.dtype)
        assert np.array_equal(have, want), key


def test_pad_len_edge_cases():
    from oracle.collate_cpu import pad_len
    assert [pad_len(n, 8) for n in (1, 8, 9, 16)] == [8, 8, 16, 16]
    assert pad_len(13, 1) == 13 and pad_len(13, 0) == 13
