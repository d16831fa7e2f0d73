"""GPU: a reference-written checkpoint restored into the device arena runs the HIP path to the reference's loss, and is written
back bit-identically (model, AdamW moments, step, hyper-parameters) in the reference's layout."""
import torch
import pytest

from test_checkpoint_cpu import CKPT, probe

pytestmark = pytest.mark.gpu


def test_reference_checkpoint_drives_the_hip_path_and_is_saved_back_bit_exactly(tmp_path):
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.checkpoint import load_checkpoint, save_checkpoint
    from scoreperformer_amd.models import ScorePerformer
    dev = torch.device("cuda")
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    model = ScorePerformer.init(ref["model"]["config"])
    arena = ParamArena(model, dev)
    opt = FusedAdamW(arena, lr=5e-2, weight_decay=0.5, grad_clip=None)
    sched = torch.optim.lr_scheduler.ExponentialLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3), gamma=0.99)
    load_checkpoint(CKPT, model, opt, lr_scheduler=sched)                       # trainer.py:389-414
    assert arena.step_count == 2 and opt.lr == ref["optimizer"]["optimizer"]["param_groups"][0]["lr"] and opt.weight_decay == 1e-2
    assert sched.last_epoch == ref["optimizer"]["lr_scheduler"]["last_epoch"]

    # the restored weights, through the HIP kernels, give the loss the reference got after saving this file
    batch, draws, loss, losses = probe()
    model.eval()
    model.perf_encoder._z_override = [z.to(dev) for z in draws]
    with torch.no_grad():
        out = model(**{k: v.to(dev) for k, v in batch.items()})
    assert abs(float(out.loss) - loss) < 1e-3 * abs(loss) + 1e-3, (float(out.loss), loss)      # north_star: loss within 1e-3
    for k, v in losses.items():
        assert abs(float(out.losses[k]) - v) < 2e-2 * max(1.0, abs(v)), k                     # per-key entries (bf16 GEMMs)

    # write it back: same layout, bit-identical tensors
    path = str(tmp_path / "resaved.pt")
    save_checkpoint(path, model, opt, model_config=ref["model"]["config"], experiment=ref["experiment"], lr_scheduler=sched)
    again = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(again) == sorted(ref) and again["experiment"] == ref["experiment"] and again["model"]["config"] == ref["model"]["config"]
    assert list(again["model"]["state_dict"]) == list(ref["model"]["state_dict"])
    for k, v in ref["model"]["state_dict"].items():
        assert torch.equal(again["model"]["state_dict"][k], v), k
    a, r = again["optimizer"]["optimizer"], ref["optimizer"]["optimizer"]
    assert a["param_groups"][0]["params"] == r["param_groups"][0]["params"]
    for key in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        assert tuple(a["param_groups"][0][key]) == tuple(r["param_groups"][0][key]) if key == "betas" else a["param_groups"][0][key] == r["param_groups"][0][key]
    assert sorted(a["state"]) == sorted(r["state"])
    for i, st in r["state"].items():
        assert float(a["state"][i]["step"]) == float(st["step"])
        assert torch.equal(a["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(a["state"][i]["exp_avg_sq"], st["exp_avg_sq"]), i
    assert again["optimizer"]["lr_scheduler"]["last_epoch"] == ref["optimizer"]["lr_scheduler"]["last_epoch"]
    # torch's own AdamW (what the reference wraps) accepts the file's optimizer entry
    topt = torch.optim.AdamW([torch.nn.Parameter(p.detach().cpu().clone()) for p in arena.param_list], lr=1.0)
    topt.load_state_dict(a)


def test_resumed_training_step_follows_the_reference_optimizer():
    """One more AdamW step from the checkpoint: fused arena update vs torch.optim.AdamW fed the SAME gradients."""
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.checkpoint import load_checkpoint
    from scoreperformer_amd.models import ScorePerformer
    dev = torch.device("cuda")
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    model = ScorePerformer.init(ref["model"]["config"])
    arena = ParamArena(model, dev)
    opt = FusedAdamW(arena, grad_clip=None)
    load_checkpoint(CKPT, model, opt)
    batch, draws, _, _ = probe()
    model.train()
    model.perf_encoder._z_override = [z.to(dev) for z in draws]
    arena.zero_grad()
    model(**{k: v.to(dev) for k, v in batch.items()}).loss.backward()
    tparams = [torch.nn.Parameter(p.detach().cpu().clone()) for p in arena.param_list]
    for tp, p in zip(tparams, arena.param_list):
        tp.grad = p.grad.detach().cpu().clone()
    topt = torch.optim.AdamW(tparams, lr=1.0)
    topt.load_state_dict(ref["optimizer"]["optimizer"])
    topt.step()
    opt.step()
    for tp, p, name in zip(tparams, arena.param_list, arena.names):
        assert torch.allclose(p.detach().cpu(), tp.detach(), rtol=1e-5, atol=1e-6), name
