"""Flat parameter arena + fused optimizer (memory laid out for one-kernel updates and bucketed RCCL all-reduce).

All unique parameters of a model are moved into ONE contiguous fp32 buffer (`params`), with parallel buffers for
gradients, Adam moments and the bf16 compute copy.  Each `nn.Parameter` keeps its name/shape (state_dict layout is
unchanged) but its storage is a view of the arena, so:

* the optimizer step (global grad-norm clip + AdamW + bf16 refresh, `experiments/optimizers.py:151-169`) is two
  launches over ~72 M contiguous floats instead of 218 tensors x several ops;
* weight-gradient GEMMs accumulate straight into `grads` (no autograd accumulation pass);
* data-parallel all-reduce works on contiguous buckets of `grads` (see parallel.py);
* q/k/v projection weights are adjacent, which gives the fused QKV GEMM operand for free.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import BF16, F32

ALIGN = 8  # elements: 16 B for bf16, 32 B for fp32


class ParamArena:
    def __init__(self, model: nn.Module, device: Optional[torch.device] = None):
        params: List[nn.Parameter] = []
        names: List[str] = []
        seen = set()
        for name, p in model.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
                names.append(name)
        device = device or params[0].device
        self.model, self.param_list, self.names, self.device = model, params, names, device
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.offsets, self.total = offsets, total
        self.params = torch.zeros(total, device=device, dtype=F32)
        self.grads = torch.zeros(total, device=device, dtype=F32)
        self.exp_avg = torch.zeros(total, device=device, dtype=F32)
        self.exp_avg_sq = torch.zeros(total, device=device, dtype=F32)
        self.shadow = torch.zeros(total, device=device, dtype=BF16)
        self._normsq = torch.zeros(1, device=device, dtype=F32)
        self._normsq_ws = None      # scratch of the fixed-order gradient norm (made on first use: the library is loaded by then)
        # torch.optim.AdamW keeps ONE STEP COUNTER PER PARAMETER (bias corrections 1 - beta^own_step): a parameter that is frozen for a
        # while, or unused in some batches (an LM-head key without valid labels), falls behind the others
        self.steps: List[int] = [0] * len(params)
        self._masks: Dict[tuple, torch.Tensor] = {}    # device slot masks by selection (a handful: they change on freeze / unfreeze)
        with torch.no_grad():
            for p, off in zip(params, offsets):
                n = p.numel()
                view = self.params[off:off + n].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=F32))
                p.data = view
                p.grad = self.grads[off:off + n].view(p.shape)
                p._spn_main_grad = p.grad
                p._spn_shadow = self.shadow[off:off + n].view(p.shape)
                p._spn_offset = off
                p._spn_touched = False
                # gradients that arrive through autograd's own accumulation (p.grad is the arena view) count as produced too
                p.register_post_accumulate_grad_hook(lambda q: setattr(q, "_spn_touched", True))
        # buffers follow the model to the device
        for mod in model.modules():
            for bname, buf in list(mod._buffers.items()):
                if buf is not None and buf.device != device:
                    mod._buffers[bname] = buf.to(device)
        self._bind_fused_groups()
        self.refresh_shadow()
        # weights written behind the optimizer's back (load_state_dict / Model.load) invalidate the bf16 compute copy
        model.register_load_state_dict_post_hook(lambda module, incompatible_keys: self.refresh_shadow())

    # -- bf16 compute copies ---------------------------------------------------------------------------------
    def refresh_shadow(self):
        """Recompute the whole bf16 copy (after load_state_dict or any out-of-band weight edit)."""
        ops.cast(self.params.view(1, -1), BF16, out=self.shadow.view(1, -1))
        for p in self.param_list:
            p._spn_shadow_version = p._version
        for mod in self.model.modules():
            for attr in getattr(mod, "_spn_fuse_groups", {}) or {}:
                fused = getattr(mod, attr, None)
                if fused is not None and hasattr(fused, "_spn_parts"):
                    fused._spn_shadow_version = tuple(p._version for p in fused._spn_parts)

    def _bind_fused_groups(self):
        by_id = {id(p): off for p, off in zip(self.param_list, self.offsets)}
        for mod in self.model.modules():
            groups = getattr(mod, "_spn_fuse_groups", None)
            if not groups:
                continue
            for attr, pnames in groups.items():
                ps = []
                for pn in pnames:
                    obj = mod
                    for part in pn.split("."):
                        obj = getattr(obj, part)
                    ps.append(obj)
                offs = [by_id.get(id(p)) for p in ps]
                ok = all(o is not None for o in offs)
                for a, b_, pa in zip(offs[:-1], offs[1:], ps[:-1]):
                    ok = ok and a is not None and b_ == a + pa.numel()   # adjacent, no padding in between
                ok = ok and all(p.shape[1:] == ps[0].shape[1:] for p in ps)
                if not ok:
                    setattr(mod, attr, None)
                    continue
                rows = sum(p.shape[0] for p in ps)
                n = rows * ps[0][0].numel()
                shape = (rows,) + tuple(ps[0].shape[1:])
                fused = self.params[offs[0]:offs[0] + n].view(shape).detach().requires_grad_(True)
                fused._spn_main_grad = self.grads[offs[0]:offs[0] + n].view(shape)
                fused._spn_shadow = self.shadow[offs[0]:offs[0] + n].view(shape)
                fused._spn_parts = ps
                fused._spn_shadow_version = tuple(p._version for p in ps)
                setattr(mod, attr, fused)

    # -- training step -----------------------------------------------------------------------------------------
    def zero_grad(self):
        self.grads.zero_()

    def grad_norm_sq(self) -> torch.Tensor:
        """Sum of squares of the whole gradient arena, summed in a FIXED order: replicas that hold the same (all-reduced) gradient get
        the same clip coefficient bit for bit and stay identical (a float-atomic sum differs in its last bit from rank to rank)."""
        self._normsq.zero_()
        if self._normsq_ws is None:
            self._normsq_ws = torch.empty(ops.sumsq_ws_floats(), device=self.grads.device, dtype=F32)
        return ops.sumsq(self.grads, out=self._normsq, ws=self._normsq_ws)

    def active_params(self) -> List[bool]:
        """Parameters torch.optim.AdamW would update now: `requires_grad` and a gradient was produced since the last step
        (the reference wraps AdamW over model.parameters(), which skips grad-is-None parameters: frozen by Model.freeze,
        models/base.py:95-102, or unused in the forward -- no weight decay, no moments; experiments/optimizers.py:151-169)."""
        touched = [bool(getattr(p, "_spn_touched", False)) for p in self.param_list]
        if getattr(self, "all_trainable_active", False):
            # data-parallel replicas (parallel.GradSync with more than one rank sets this): "was a gradient produced" is a LOCAL fact -- a
            # key without valid labels in this rank's batch -- while the all-reduced gradient is the same everywhere; replicas that
            # disagree on which parameters to step diverge.  Every trainable parameter is stepped with the reduced gradient (zero where
            # no rank produced one: weight decay and moment decay only), with no flag exchange and no host read.
            touched = [True] * len(touched)
        if not any(touched):   # gradients written into the arena out of band (no backward ran): every trainable parameter has one
            touched = [True] * len(touched)
        return [bool(p.requires_grad and t) for p, t in zip(self.param_list, touched)]

    @property
    def step_count(self) -> int:
        """Steps taken by the most-stepped parameter (every parameter's own count: `steps`)."""
        return max(self.steps) if self.steps else 0

    @property
    def updated(self) -> List[bool]:
        return [s > 0 for s in self.steps]

    def _mask_for(self, active: List[bool]) -> Optional[torch.Tensor]:
        if all(active):
            return None
        key = tuple(active)
        m = self._masks.get(key)
        if m is None:   # built only when a selection is new (freeze / unfreeze, a key without labels), not every step
            host = torch.zeros(self.total // ALIGN, dtype=torch.uint8)
            bounds = self.offsets[1:] + [self.total]
            for on, s, e in zip(active, self.offsets, bounds):
                if on:
                    host[s // ALIGN:e // ALIGN] = 1
            if len(self._masks) >= 16:
                self._masks.pop(next(iter(self._masks)))
            m = self._masks[key] = host.to(self.device)
        return m

    def step(self, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
             max_norm: Optional[float] = None, grad_scale: float = 1.0) -> torch.Tensor:
        """clip_grad_norm_(max_norm) + AdamW over the arena; returns the (unclipped) grad norm as a device scalar.
        Parameters without a gradient this step (see `active_params`) are left untouched, as torch.optim.AdamW leaves them, and every
        parameter is bias-corrected with ITS OWN step count: one launch when all updated parameters share it (always, unless parameters
        were frozen or unused for some steps), else one launch per distinct count over that count's slots."""
        active = self.active_params()
        for i, on in enumerate(active):
            if on:
                self.steps[i] += 1
        normsq = self.grad_norm_sq()
        for sv in sorted({s for s, on in zip(self.steps, active) if on}):
            sel = [on and s == sv for s, on in zip(self.steps, active)]
            ops.adamw_step(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.shadow, normsq,
                           max_norm=max_norm or 0.0, grad_scale=grad_scale, lr=lr, betas=betas, eps=eps,
                           weight_decay=weight_decay, step=sv, slot_mask=self._mask_for(sel))
        for p in self.param_list:
            p._spn_touched = False
        return normsq.sqrt() * grad_scale

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": torch.tensor(self.step_count),
                "steps": torch.tensor(self.steps, dtype=torch.int64)}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.steps = [int(v) for v in sd["steps"]] if "steps" in sd else [int(sd["step"])] * len(self.param_list)


class FusedAdamW:
    """Optimizer facade with the reference's hyper-parameters (recipes/default.yaml:79-89): AdamW(lr 2e-4, wd 1e-6),
    grad clip 2.0; `step()` = unscale/clip/update/zero_grad of `Optimizer.step` (experiments/optimizers.py:151-169)."""

    def __init__(self, arena: ParamArena, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-6, grad_clip: Optional[float] = 2.0):
        self.arena, self.lr, self.betas, self.eps, self.weight_decay, self.grad_clip = arena, lr, betas, eps, weight_decay, grad_clip

    def step(self, grad_scale: float = 1.0, zero_grad: bool = True):
        norm = self.arena.step(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                               max_norm=self.grad_clip, grad_scale=grad_scale)
        if zero_grad:
            self.arena.zero_grad()
        return norm

    # -- checkpoint / resume in torch.optim.AdamW's own format (what the reference's checkpoints hold under "optimizer":
    #    experiments/optimizers.py wraps torch.optim.AdamW over model.parameters(), i.e. the arena's parameter order) ------------
    def state_dict(self):
        a = self.arena
        state = {}
        if a.step_count > 0:
            for i, (p, off) in enumerate(zip(a.param_list, a.offsets)):
                if a.steps[i] == 0:
                    continue   # torch.optim.AdamW holds no state for a parameter it never stepped (frozen / unused)
                n = p.numel()
                state[i] = {"step": torch.tensor(float(a.steps[i])),
                            "exp_avg": a.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": a.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(a.param_list)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        a = self.arena
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(a.param_list):
            raise ValueError("optimizer state does not match the model: expected one param group with "
                             f"{len(a.param_list)} parameters")
        g = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        a.exp_avg.zero_(); a.exp_avg_sq.zero_()
        steps = [0] * len(a.param_list)
        for i, (p, off) in enumerate(zip(a.param_list, a.offsets)):
            st = sd["state"].get(g["params"][i])
            if st is None:
                continue
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {a.names[i]} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
            a.exp_avg[off:off + n].view(p.shape).copy_(st["exp_avg"])
            a.exp_avg_sq[off:off + n].view(p.shape).copy_(st["exp_avg_sq"])
            steps[i] = int(st["step"])   # per parameter, as torch.optim.AdamW keeps them (they differ after a freeze / unfreeze)
        a.steps = steps
