"""CPU: the oracle on the second variant set (oracle/variants.py NAMES2: lm-tied-split head, absolute positions, GELU / un-gated
feed-forwards, regression head + L1 loss, decoder-only Performer in CLM / MLM / MixedLM mode) against the REFERENCE's own loss, loss
dict and per-parameter gradient norms (tests/golden/variants2.npz, written by oracle/refimport/make_golden_variants2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.variants import NAMES2, SMALL_VOCAB, variant2_config, variant2_inputs
from oracle.weights import filled_state_dict

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "variants2.npz"))


def build(name):
    from scoreperformer_amd.models import Performer, ScorePerformer
    cls = Performer if name.startswith("performer_") else ScorePerformer
    return cls.init(variant2_config(name))


def oracle_forward(name, sd, batch):
    cfg = variant2_config(name)
    if name.startswith("performer_"):
        return ref_cpu.performer_forward(sd, cfg, variant2_inputs(name, batch))
    z = [torch.from_numpy(Z[f"{name}/z{i}"]) for i in range(len(cfg["perf_encoder"]["latent_dim"]))]
    return ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True)


@pytest.mark.parametrize("name", NAMES2)
def test_oracle_matches_the_reference_on_variant_set_2(name):
    from scoreperformer_amd.synthetic import synthetic_batch
    batch = synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)
    sd = filled_state_dict(build(name), seed=1)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v) for k, v in sd.items()}
    # tied tensors share one leaf so that gradients accumulate like in the reference module tree
    from oracle.weights import canonical
    leaves = {}
    for k in list(sd):
        sd[k] = leaves.setdefault(canonical(k), sd[k])
    res = oracle_forward(name, sd, batch)
    assert abs(float(res["loss"]) - float(Z[f"{name}/loss"])) < 2e-5, (float(res["loss"]), float(Z[f"{name}/loss"]))
    want = {k.split("/losses/", 1)[1]: float(Z[k]) for k in Z.files if k.startswith(f"{name}/losses/")}
    assert set(want) == set(res["losses"]), (sorted(want), sorted(res["losses"]))
    for k, v in want.items():
        assert abs(float(res["losses"][k]) - v) < 2e-5, (k, float(res["losses"][k]), v)
    res["loss"].backward()
    checked = 0
    for k in Z.files:
        if not k.startswith(f"{name}/gradnorm/"):
            continue
        pname = k.split("/gradnorm/", 1)[1]
        g = sd[pname].grad
        assert g is not None, pname
        got, ref = float(g.norm()), float(Z[k])
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-3) + 1e-7, (pname, got, ref)
        checked += 1
    assert checked >= 50
