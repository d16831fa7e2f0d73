"""CPU: the oracle on the third variant set (oracle/variants.py NAMES3: `multiseq_mode="pre-sum"` and `TupleTokenEmbeddings(mode="sum")`,
the reference's own DEFAULTS, models/scoreperformer/embeddings.py:66-69,117,141,171,231-241) against the REFERENCE's loss, loss dict
and per-parameter gradient norms (tests/golden/variants3.npz, written by oracle/refimport/make_golden_variants3.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.variants import NAMES3, SMALL_VOCAB, variant3_config
from oracle.weights import filled_state_dict

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "variants3.npz"))


@pytest.mark.parametrize("name", NAMES3)
def test_oracle_matches_the_reference_on_variant_set_3(name):
    from oracle.weights import canonical
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import synthetic_batch
    batch = synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)
    cfg = variant3_config(name)
    model = ScorePerformer.init(variant3_config(name))
    if name == "emb_mode_sum":      # `sum` mode: one common width, no projection in front of the transformer (embeddings.py:117,141)
        assert model.perf_decoder.model.token_emb.total_emb_dim == int(cfg["dim"])
        assert not hasattr(model.perf_decoder.model.token_emb, "project_emb")
    sd = filled_state_dict(model, seed=1)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v) for k, v in sd.items()}
    leaves = {}
    for k in list(sd):
        sd[k] = leaves.setdefault(canonical(k), sd[k])
    z = [torch.from_numpy(Z[f"{name}/z{i}"]) for i in range(len(cfg["perf_encoder"]["latent_dim"]))]
    res = ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True)
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
