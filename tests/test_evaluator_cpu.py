"""CPU: the evaluator oracle (oracle/evaluator_cpu.py) against the reference evaluator's own outputs (tests/golden/evaluator.npz)."""
import ast
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "evaluator.npz")


def load_cases():
    z = np.load(GOLD, allow_pickle=False)
    cases = {}
    for name in sorted({k.split("/")[0] for k in z.files}):
        cfg = ast.literal_eval(str(z[f"{name}/cfg"]))
        keys = [k.split("/logits/")[1] for k in z.files if k.startswith(f"{name}/logits/")]
        cases[name] = dict(cfg=cfg, labels=z[f"{name}/labels"], logits={k: z[f"{name}/logits/{k}"] for k in keys},
                           values={k: z[f"{name}/values/{k}"] for k in keys},
                           metrics={k.split("/metric/")[1]: float(z[k]) for k in z.files if k.startswith(f"{name}/metric/")})
    return cases


CASES = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_evaluator_matches_the_reference(name):
    from oracle.evaluator_cpu import evaluate
    c = CASES[name]
    cfg = c["cfg"]
    got = evaluate(c["logits"], c["labels"], shift=cfg["mode"] in ("clm", "mixlm"), ignore_keys=cfg["ignore_keys"],
                   token_values=c["values"] if cfg["with_values"] else None, weighted_distance=cfg["weighted"])
    assert sorted(got) == sorted(c["metrics"])            # same metric names: the "has any valid label" / ignore-key gating
    for k, want in c["metrics"].items():
        assert abs(got[k] - want) <= 2e-6 * max(1.0, abs(want)), (k, got[k], want)   # the reference accumulates in fp32
