"""CPU: the model oracle (oracle/ref_cpu.py) on the reference's ablation-recipe model variants, against the reference's own losses
(tests/golden/ablations.npz, oracle/refimport/make_golden_ablations.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.variants import NAMES, SMALL_VOCAB, ablation_config, variant_batch

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ablations.npz"))


def golden(name):
    draws = [torch.from_numpy(Z[f"{name}/z{i}"]) for i in range(len([k for k in Z.files if k.startswith(f"{name}/z")]))]
    losses = {k.split("/losses/")[1]: float(Z[k]) for k in Z.files if k.startswith(f"{name}/losses/")}
    return draws, float(Z[f"{name}/loss"]), losses


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_on_ablation_variants(name):
    from oracle import ref_cpu
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import synthetic_batch
    cfg = ablation_config(name)
    sd = filled_state_dict(ScorePerformer.init(ablation_config(name)), seed=1)
    batch = variant_batch(name, synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB))
    draws, loss, losses = golden(name)
    with torch.no_grad():
        got = ref_cpu.score_performer_forward(sd, cfg, batch, draws, training=True)
    assert abs(float(got["loss"]) - loss) < 2e-5 * abs(loss), (float(got["loss"]), loss)
    assert sorted(got["losses"]) == sorted(losses)
    for k, v in losses.items():
        assert abs(float(got["losses"][k]) - v) < 2e-5 * max(1.0, abs(v)), k
