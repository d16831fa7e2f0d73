"""Checkpoint I/O in the reference's own file layout.

`experiments/trainer.py:296-314` writes, `trainer.py:389-414` + `models/base.py:43-93` read:

    {"experiment": {"config", "trainer", "state"},
     "model": {"config": <plain dict>, "state_dict": {name: tensor}},
     "optimizer": {"optimizer": <torch.optim.AdamW state_dict>, "lr_scheduler": {...}}}       # absent when minimal

so a file written here loads in the reference (`Model.from_pretrained`, `Trainer.load_checkpoint`) and a file written by the reference
loads here.  Parameters live in one flat fp32 device arena (`arena.ParamArena`): saving moves the arena (and the two AdamW moment
arenas) to the host with ONE copy each and slices the per-parameter tensors from the host image, instead of one D2H copy per
parameter; loading writes through `load_state_dict`, after which the arena refreshes its bf16 compute copy.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import torch

from .utils.config import OmegaConf

logger = logging.getLogger("scoreperformer_amd")


def model_state_dict(model, arena=None) -> Dict[str, torch.Tensor]:
    """`model.state_dict()` on the host, each tensor with its own storage (what `torch.save` of the reference's model holds)."""
    if arena is None:
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = arena.params.cpu()                                             # one D2H copy of every parameter
    by_id = {id(p): (off, p) for p, off in zip(arena.param_list, arena.offsets)}
    named = {name: p for name, p in model.named_parameters(remove_duplicate=False)}
    out = {}
    for key, value in model.state_dict().items():
        p = named.get(key)
        if p is not None and id(p) in by_id:
            off, _ = by_id[id(p)]
            out[key] = host[off:off + p.numel()].view(p.shape).clone()
        else:                                                             # buffers
            out[key] = value.detach().cpu().clone()
    return out


def optimizer_state_dict(optimizer, lr_scheduler=None) -> Dict[str, object]:
    """The reference's `Optimizer.state_dict()` (experiments/optimizers.py:195-199)."""
    sd = {"optimizer": _adamw_state_on_host(optimizer)}
    if lr_scheduler is not None:
        sd["lr_scheduler"] = lr_scheduler.state_dict() if hasattr(lr_scheduler, "state_dict") else dict(lr_scheduler)
    return sd


def _adamw_state_on_host(optimizer):
    a = optimizer.arena
    sd = optimizer.state_dict()
    if a.step_count > 0:                                                   # two bulk copies instead of 2 per parameter
        m, v = a.exp_avg.cpu(), a.exp_avg_sq.cpu()
        for i, (p, off) in enumerate(zip(a.param_list, a.offsets)):
            if i not in sd["state"]:
                continue                                                   # never stepped: torch.optim.AdamW holds no state for it
            n = p.numel()
            sd["state"][i] = {"step": torch.tensor(float(a.steps[i])), "exp_avg": m[off:off + n].view(p.shape).clone(),
                              "exp_avg_sq": v[off:off + n].view(p.shape).clone()}
    return sd


def _plain(obj):
    """Config containers -> builtin dicts / lists (a checkpoint must unpickle without this package)."""
    if obj is None or isinstance(obj, (str, int, float, bool)):
        return obj
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if hasattr(obj, "items"):
        return {str(k): _plain(v) for k, v in obj.items()}
    return OmegaConf.to_container(obj, resolve=True) if hasattr(OmegaConf, "to_container") else obj


def save_checkpoint(path: str, model, optimizer=None, *, model_config=None, experiment: Optional[dict] = None, lr_scheduler=None,
                    minimal: bool = False) -> dict:
    """`Trainer._save_checkpoint` (trainer.py:296-314).  `model_config`: dict / attr-dict of the model section; `experiment`:
    {"config", "trainer", "state"} JSON strings (kept verbatim)."""
    cfg = _plain(model_config)   # plain dict / list / scalars, like OmegaConf.to_container(..., resolve=True) in the reference
    checkpoint = {
        "experiment": dict({"config": None, "trainer": None, "state": None}, **(experiment or {})),
        "model": {"config": cfg, "state_dict": model_state_dict(model, getattr(optimizer, "arena", None))},
    }
    if not minimal and optimizer is not None:
        checkpoint["optimizer"] = optimizer_state_dict(optimizer, lr_scheduler)
    logger.info(f"*** Saving checkpoint {path} ***")
    torch.save(checkpoint, path)
    return checkpoint


def load_checkpoint(path: str, model, optimizer=None, *, warm_start: bool = False, ignore_layers: Optional[List[str]] = None,
                    ignore_mismatched_keys: bool = False, lr_scheduler=None, restore_lr: bool = True) -> dict:
    """`Trainer.load_checkpoint` (trainer.py:389-414): warm start = tolerant `Model.load`; otherwise strict-shaped load plus the
    optimizer (and, with `restore_lr`, the scheduler).  Returns the checkpoint dict."""
    logger.info(f"*** Loading checkpoint `{path}` ***")
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    state = checkpoint["model"]["state_dict"]
    if warm_start:
        model.load(state, ignore_layers, ignore_mismatched_keys)
    else:
        model.load(state, None, False)
        if "optimizer" in checkpoint and optimizer is not None:
            osd = checkpoint["optimizer"]
            optimizer.load_state_dict(osd["optimizer"] if "optimizer" in osd else osd)
            if lr_scheduler is not None:
                if restore_lr and osd.get("lr_scheduler") is not None:
                    lr_scheduler.load_state_dict(osd["lr_scheduler"])
                elif not restore_lr:
                    optimizer.lr = lr_scheduler.get_last_lr()[0]          # optimizers.py:190-193
    return checkpoint
