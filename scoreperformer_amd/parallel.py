"""Data-parallel gradient synchronisation over RCCL/xGMI (one process per GPU, `torch.distributed` backend "nccl").

The reference is single-device; DP semantics = its own gradient-accumulation semantics per rank.
Gradients live in ONE contiguous fp32 arena, so all-reduce works on contiguous buckets.  A bucket is launched
(asynchronously, on RCCL's stream) as soon as every parameter in it has received its last gradient contribution of the
backward pass, i.e. while the rest of backward is still running; the 1/world factor is folded into the optimizer kernel.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, arena, group=None, bucket_mb: float = 32.0, force: bool = False, grad_dtype: torch.dtype = torch.float32,
                 dry_run: bool = False, transport: str = "torch", persistent_backward: Optional[bool] = None):
        """`force=True` keeps the bucketed all-reduce path active on a one-rank group (single-GPU tests of the RCCL path).
        `grad_dtype=torch.bfloat16` sends every bucket as bf16 (half the bytes over xGMI: 144 MB instead of 288 MB per step at C3;
        the sum over ranks is then formed in bf16 -- about 3 significant digits per element -- so fp32 stays the default).
        `dry_run=True` runs the whole readiness protocol without a process group: buckets are "launched" into `self.events`
        (tools/record_grad_events.py records the real model's pending / ready order that way).
        `transport="spn"`: buckets go through libspn.so's own RCCL wrapper (`spn_comm_allreduce`: ncclAllReduce on a dedicated
        communication stream, event-fenced; comm.NativeComm, created collectively over `group`) instead of `dist.all_reduce`;
        it needs a process group and raises without one (no silent fall-back to the torch path).
        `persistent_backward`: the process-wide `gemm_persist_bwd` knob (csrc/tuning.h) -- input-gradient GEMMs walk their tiles with
        one persistent block per CU, which is only safe when NO other kernel holds CUs during the backward (a concurrent all-reduce
        starves the blocks that land on its CUs).  An ACTIVE object (it launches all-reduces from inside backward) always forces the knob
        to 0, whatever this argument says and whatever an earlier object left behind.  Otherwise: None (default) leaves the knob alone;
        True asks for the walk -- the caller vouches that nobody else reduces during backward either; False switches it off.  `close()`
        (also the context manager's exit and `__del__`) restores the value found at construction."""
        if transport not in ("torch", "spn"):
            raise ValueError(f"unknown transport {transport!r}")
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.dry_run = dry_run
        self.active = dry_run or self.world > 1 or (force and group is not None)
        self.grad_dtype = grad_dtype
        self.native = None
        if transport == "spn" and not dry_run:
            if group is None:
                raise ValueError("GradSync(transport='spn') needs a process group (the RCCL id travels over it)")
            if self.world > 1 or force:
                from .comm import NativeComm
                self.native = NativeComm.from_group(group)
        self._knob_before = None
        self._knob_set = None
        self._hooked: List = []         # (object, attribute) pairs and hook handles this object installed: close() removes them
        on_gpu = bool(getattr(getattr(arena, "grads", None), "is_cuda", False))    # (the gloo tests drive this class with host tensors)
        if not dry_run and (persistent_backward is not None or (self.active and on_gpu)):
            from . import lib
            self._knob_before = lib.get_tuning("gemm_persist_bwd")
            self._knob_set = 1.0 if (persistent_backward and not self.active) else 0.0
            lib.set_tuning("gemm_persist_bwd", self._knob_set)
        self._counts_as_dp = bool(self.active and self.world > 1 and not dry_run)
        if self._counts_as_dp:   # see ParamArena.active_params: replicas must agree on the set of parameters they step
            arena._dp_syncs = getattr(arena, "_dp_syncs", 0) + 1     # (a count: two objects on one arena may overlap in time)
            arena.all_trainable_active = True
        self.handles: List = []
        self.buckets: List[tuple] = []   # (start, end) element ranges of arena.grads
        self.bucket_of = {}
        self.pending: List[int] = []
        self.events: List[tuple] = []    # dry_run: ("pending" | "ready" | "autograd" | "launch", index) in program order
        if not self.active:
            return
        # buckets in REVERSE arena order (decoder parameters come last in the arena and first in backward)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        bounds = list(zip(arena.offsets, arena.offsets[1:] + [arena.total]))
        cur_end, cur_start, members, groups = None, None, [], []
        for idx in reversed(range(len(bounds))):
            s, e = bounds[idx]
            if cur_end is None:
                cur_end = e
            members.append(idx)
            cur_start = s
            if cur_end - cur_start >= cap:
                groups.append((cur_start, cur_end, members))
                cur_end, members = None, []
        if members:
            groups.append((cur_start, cur_end, members))
        for b, (s, e, mem) in enumerate(groups):
            self.buckets.append((s, e))
            for idx in mem:
                self.bucket_of[idx] = b
        self._install_hooks()

    # every gradient producer calls p._spn_grad_pending() in forward and p._spn_grad_ready() once its contribution has
    # been enqueued in backward (functional.py); parameters whose gradient arrives through autograd's own accumulation
    # use a post-accumulate hook.
    def _install_hooks(self):
        self.counts = [0] * len(self.arena.param_list)
        self.bucket_left = [0] * len(self.buckets)
        for idx, p in enumerate(self.arena.param_list):
            p._spn_grad_pending = (lambda i=idx: self._pending(i))
            p._spn_grad_ready = (lambda i=idx: self._ready(i))
            self._hooked.append(p)
            self._hooked.append(p.register_post_accumulate_grad_hook(lambda _p, i=idx: self._autograd_ready(i)))
        for mod in self.arena.model.modules():
            for attr in getattr(mod, "_spn_fuse_groups", {}) or {}:
                fused = getattr(mod, attr, None)
                if fused is not None and hasattr(fused, "_spn_parts"):
                    idxs = [self._index_of(p) for p in fused._spn_parts]
                    fused._spn_grad_pending = (lambda ii=idxs: [self._pending(i) for i in ii])
                    fused._spn_grad_ready = (lambda ii=idxs: [self._ready(i) for i in ii])
                    self._hooked.append(fused)

    def _index_of(self, p):
        for i, q in enumerate(self.arena.param_list):
            if q is p:
                return i
        raise KeyError

    def _pending(self, i):
        if self.dry_run:
            self.events.append(("pending", i))
        self.counts[i] += 1

    def _ready(self, i):
        if self.dry_run:
            self.events.append(("ready", i))
        self.counts[i] -= 1
        if self.counts[i] == 0:
            self._param_done(i)

    def _autograd_ready(self, i):
        # autograd accumulated a returned gradient into p.grad; such parameters are not counted in forward
        if self.dry_run:
            self.events.append(("autograd", i))
        if self.counts[i] == 0:
            self._param_done(i)

    def _param_done(self, i):
        if i in self.done:
            return
        self.done.add(i)
        b = self.bucket_of[i]
        self.bucket_left[b] -= 1
        if self.bucket_left[b] == 0:
            self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        self.launched.add(b)
        if self.dry_run:
            self.events.append(("launch", b))
            return
        grads = self.arena.grads[s:e]
        if self.grad_dtype == grads.dtype:
            if self.native is not None:
                self.native.all_reduce_(grads)
                self.handles.append((None, None, None))
            else:
                self.handles.append((dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
            return
        # reduced-precision transport: the bucket is cast into a staging buffer (stream-ordered behind the kernels that produced
        # the gradients), reduced there, and cast back into the fp32 arena in finish()
        staged = self._staging(b, e - s, grads.device)
        staged.copy_(grads)
        if self.native is not None:
            self.native.all_reduce_(staged)
            self.handles.append((None, staged, grads))
        else:
            self.handles.append((dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=self.group, async_op=True), staged, grads))

    def _staging(self, b, n, device):
        bufs = self.__dict__.setdefault("_stage_bufs", {})
        if b not in bufs:
            bufs[b] = torch.empty(n, dtype=self.grad_dtype, device=device)
        return bufs[b]

    def close(self):
        """Restore the `gemm_persist_bwd` knob this object changed (if it did, and only while the knob still holds the value this object
        wrote: a later object that set it again owns it now, and a late `close()` / garbage collection of this one must not re-enable the
        persistent walk under the other's all-reduces), take this object's gradient hooks off the parameters (the arena can then be used
        without it, or with a new object) and drop the native communicator."""
        if self._knob_before is not None:
            from . import lib
            if lib.get_tuning("gemm_persist_bwd") == self._knob_set:
                lib.set_tuning("gemm_persist_bwd", self._knob_before)
            self._knob_before = None
        for h in self._hooked:
            if hasattr(h, "remove"):
                h.remove()
            else:
                for attr in ("_spn_grad_pending", "_spn_grad_ready"):
                    if attr in getattr(h, "__dict__", {}):
                        delattr(h, attr)
        self._hooked = []
        if self._counts_as_dp:
            self._counts_as_dp = False
            self.arena._dp_syncs = max(0, getattr(self.arena, "_dp_syncs", 1) - 1)
            self.arena.all_trainable_active = self.arena._dp_syncs > 0
        self.active = False
        if self.native is not None:
            self.native.close()
            self.native = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if self._knob_before is not None:
                self.close()
        except BaseException:   # noqa: BLE001 -- interpreter teardown: the binding may be gone
            pass

    def begin_step(self):
        if not self.active:
            return
        if self._knob_set == 0.0 and self._knob_before is not None:
            from . import lib
            lib.set_tuning("gemm_persist_bwd", 0.0)     # re-asserted every step: nothing may switch the persistent walk on under an all-reduce
        self.handles, self.done, self.launched, self.events = [], set(), set(), []
        self.counts = [0] * len(self.arena.param_list)
        left = [0] * len(self.buckets)
        for i in range(len(self.arena.param_list)):
            left[self.bucket_of[i]] += 1
        self.bucket_left = left

    def finish(self):
        """After backward: reduce whatever has not been launched yet (parameters that got no gradient), wait for all."""
        if not self.active:
            return
        for b in range(len(self.buckets)):
            if b not in self.launched:
                self._launch(b)
        if self.native is not None:
            self.native.wait()   # the current stream waits (on the device) for every bucket; no host synchronisation
        for h, staged, grads in self.handles:
            if h is not None:
                h.wait()
            if staged is not None:
                grads.copy_(staged)
        self.handles = []
