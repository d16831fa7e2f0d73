"""GPU: the HIP path on the third variant set (oracle/variants.py NAMES3: multi-sequence `pre-sum` token embeddings and the `sum`
embedding mode -- the reference's defaults, models/scoreperformer/embeddings.py:66-69,141,171,231-241) against the REFERENCE's own loss,
loss dict and per-parameter gradient norms (tests/golden/variants3.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "variants3.npz"))


@pytest.mark.parametrize("name", ["multiseq_pre_sum", "emb_mode_sum", "hier_no_context", "agg_same", "isolated_bar"])
def test_variant_set_3_matches_the_reference(dev, name):
    from oracle.variants import SMALL_VOCAB, variant3_config
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import synthetic_batch
    cfg = variant3_config(name)
    model = ScorePerformer.init(variant3_config(name))
    model.load_state_dict(filled_state_dict(model, seed=1), strict=True)
    arena = ParamArena(model, dev)
    model.train()
    model.perf_encoder._z_override = [torch.from_numpy(Z[f"{name}/z{i}"]).to(dev) for i in range(len(cfg["perf_encoder"]["latent_dim"]))]
    batch = synthetic_batch(2, 40, seed=5, ragged=True, num_tokens=SMALL_VOCAB)
    out = model(**{k: v.to(dev) for k, v in batch.items()})
    loss = float(Z[f"{name}/loss"])
    want = {k.split("/losses/", 1)[1]: float(Z[k]) for k in Z.files if k.startswith(f"{name}/losses/")}
    assert set(out.losses) == set(want), (sorted(out.losses), sorted(want))
    assert abs(float(out.loss.detach()) - loss) <= 1e-2 * abs(loss), (float(out.loss.detach()), loss)   # tiny model, bf16 GEMMs
    for k, v in want.items():
        assert abs(float(out.losses[k]) - v) <= 2e-2 * max(1.0, abs(v)), (k, float(out.losses[k]), v)
    arena.zero_grad()
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    bad, checked = [], 0
    for k in Z.files:
        if k.startswith(f"{name}/gradnorm/"):
            pname = k.split("/gradnorm/", 1)[1]
            got, ref = float(named[pname].grad.double().norm()), float(Z[k])
            checked += 1
            if abs(got - ref) > 0.06 * ref + 2e-3:
                bad.append((pname, got, ref))
    assert checked >= 50 and not bad, bad[:10]
