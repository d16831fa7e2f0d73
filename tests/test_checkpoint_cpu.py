"""CPU: checkpoint I/O against a checkpoint file written by the reference itself (tests/golden/micro_checkpoint.pt)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CKPT = os.path.join(GOLD, "micro_checkpoint.pt")


def probe():
    z = np.load(os.path.join(GOLD, "micro_checkpoint_probe.npz"))
    batch = {k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch/")}
    draws = [torch.from_numpy(z[f"z{i}"]) for i in range(len([k for k in z.files if k.startswith("z")]))]
    losses = {k.split("/", 1)[1]: float(z[k]) for k in z.files if k.startswith("loss/")}
    return batch, draws, float(z["loss"]), losses


def test_reference_checkpoint_loads_strictly_and_round_trips(tmp_path):
    from scoreperformer_amd.checkpoint import save_checkpoint
    from scoreperformer_amd.models import ScorePerformer
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    model = ScorePerformer.from_pretrained(CKPT)                       # models/base.py:43-53: config -> init -> strict load
    sd = model.state_dict()
    assert list(sd) == list(ref["model"]["state_dict"])                # same keys, same order
    for k, v in ref["model"]["state_dict"].items():
        assert sd[k].dtype == v.dtype and torch.equal(sd[k], v), k
    out = str(tmp_path / "again.pt")
    save_checkpoint(out, model, None, model_config=ref["model"]["config"], experiment=ref["experiment"], minimal=True)
    again = torch.load(out, map_location="cpu", weights_only=False)
    assert sorted(again) == ["experiment", "model"] and again["experiment"] == ref["experiment"]
    assert again["model"]["config"] == ref["model"]["config"]
    assert list(again["model"]["state_dict"]) == list(ref["model"]["state_dict"])
    for k, v in ref["model"]["state_dict"].items():
        assert torch.equal(again["model"]["state_dict"][k], v), k


def test_oracle_reproduces_the_reference_loss_from_the_checkpoint():
    from oracle import ref_cpu
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    batch, draws, loss, losses = probe()
    got = ref_cpu.score_performer_forward(ref["model"]["state_dict"], ref["model"]["config"], batch, draws, training=False)
    assert abs(float(got["loss"]) - loss) < 1e-5 * abs(loss)
    for k, v in losses.items():
        assert abs(float(got["losses"][k]) - v) < 1e-5 * max(1.0, abs(v)), k


def test_warm_start_ignores_layers_and_mismatched_keys():
    from scoreperformer_amd.models import ScorePerformer
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    model = ScorePerformer.init(ref["model"]["config"])
    before = {k: v.clone() for k, v in model.state_dict().items()}
    state = dict(ref["model"]["state_dict"])
    state["not.in.the.model"] = torch.zeros(3)
    bad = next(k for k in state if k.endswith("to_out.weight"))
    state[bad] = torch.zeros(5, 7)                                     # wrong shape -> skipped with ignore_mismatched_keys
    model.load(state, ignore_layers=["score_encoder"], ignore_mismatched_keys=True)     # models/base.py:55-93
    after = model.state_dict()
    for k in after:
        if "score_encoder.token_emb" in k:
            continue                                                    # tables shared with the other stacks load under their names
        if "score_encoder" in k or k == bad:
            assert torch.equal(after[k], before[k]), k
        else:
            assert torch.equal(after[k], ref["model"]["state_dict"][k]), k


def test_saved_config_is_plain_builtins(tmp_path):
    """The model config goes into the file as builtin dicts / lists (trainer.py:305: OmegaConf.to_container), whatever container
    the caller holds: the file must unpickle without this package's config classes."""
    import pickle
    from scoreperformer_amd.checkpoint import save_checkpoint
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config
    from oracle.variants import SMALL_VOCAB
    cfg = model_config(preset="tiny", num_tokens=SMALL_VOCAB, dim=32, heads=2, depths=(1, 1, 1), emb_dims=8, latent_dim=[8, 4, 2, 2], max_seq_len=64)
    model = ScorePerformer.init(cfg)                      # `init` wraps the dict in the package's attr-dict container
    out = str(tmp_path / "c.pt")
    save_checkpoint(out, model, None, model_config=model.config if hasattr(model, "config") else cfg, minimal=True)
    again = torch.load(out, map_location="cpu", weights_only=True)     # weights_only: builtin containers and tensors only

    def check(o):
        assert type(o) in (dict, list, str, int, float, bool, type(None)), type(o)
        for v in (o.values() if isinstance(o, dict) else o if isinstance(o, list) else ()):
            check(v)
    check(again["model"]["config"])
    assert ScorePerformer.from_pretrained(out) is not None
