"""CPU (gloo, world_size 2): the bucketed gradient synchronisation reduces every arena element exactly once, launches
buckets as soon as their parameters are ready, and is insensitive to the order in which gradients arrive."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeArena:
    """CPU stand-in with the attributes GradSync reads (the real arena needs the GPU for its bf16 copies)."""

    def __init__(self, sizes):
        self.model = torch.nn.Module()
        self.param_list = [torch.nn.Parameter(torch.zeros(s)) for s in sizes]
        self.offsets, total = [], 0
        for s in sizes:
            self.offsets.append(total)
            total += (s + 7) // 8 * 8
        self.total = total
        self.grads = torch.zeros(total)
        for p, off in zip(self.param_list, self.offsets):
            p.grad = self.grads[off:off + p.numel()].view(p.shape)
            p._spn_main_grad = p.grad


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, order, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scoreperformer_amd.parallel import GradSync
    sizes = [1000, 24, 4096, 8, 300, 2048, 77]
    arena = FakeArena(sizes)
    sync = GradSync(arena, dist.group.WORLD, bucket_mb=0.004)   # ~1k floats per bucket -> several buckets
    assert len(sync.buckets) >= 3
    for step in range(2):
        sync.begin_step()
        arena.grads.zero_()
        for i in order:               # "forward": every producer announces a pending contribution
            arena.param_list[i]._spn_grad_pending()
        launched_before_finish = 0
        for i in reversed(order):     # "backward": contributions land, hooks fire
            p = arena.param_list[i]
            p.grad += (rank + 1) * (i + 1) + step
            p._spn_grad_ready()
            launched_before_finish = len(sync.launched)
        assert launched_before_finish >= len(sync.buckets) - 1   # overlap: buckets go out during "backward"
        sync.finish()
        for i, p in enumerate(arena.param_list):
            want = sum((r + 1) * (i + 1) + step for r in range(world))
            assert torch.all(p.grad == want), (i, p.grad.flatten()[:3], want)
    q.put((rank, "ok"))
    dist.destroy_process_group()


@pytest.mark.parametrize("order", [[0, 1, 2, 3, 4, 5, 6], [3, 0, 6, 2, 5, 1, 4]])
def test_gradsync_gloo_world2(order):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, order, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


# ---- the REAL model's protocol, replayed ----------------------------------------------------------------------------------------
# tests/golden/grad_events.json (tools/record_grad_events.py, recorded on MI355X) holds, for two model variants, the arena layout and
# the exact order in which the model's autograd functions call _spn_grad_pending (forward) and _spn_grad_ready / the autograd
# accumulation hook (backward) -- several contributions per tied table, fused q|k|v groups signalling three parameters at once.

class RecordedArena(FakeArena):
    def __init__(self, rec):
        self.model = torch.nn.Module()
        self.param_list = [torch.nn.Parameter(torch.zeros(s)) for s in rec["sizes"]]
        self.offsets, self.total = list(rec["offsets"]), rec["total"]
        self.grads = torch.zeros(self.total)
        for p, off in zip(self.param_list, self.offsets):
            p.grad = self.grads[off:off + p.numel()].view(p.shape)
            p._spn_main_grad = p.grad


def _replay_worker(rank, world, port, variant, grad_dtype, q):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scoreperformer_amd.parallel import GradSync
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grad_events.json")))[variant]
    arena = RecordedArena(rec)
    sync = GradSync(arena, dist.group.WORLD, bucket_mb=0.25, grad_dtype=getattr(torch, grad_dtype))
    assert len(sync.buckets) >= 4
    # contributions: every `ready` (a kernel accumulated straight into the arena), and `autograd` for parameters that never announce
    # one (their gradient arrives through autograd's own accumulation: table MLPs, index rows, ALiBi log-slopes).  For announced
    # parameters the accumulation hook fires as well, after their last `ready`, carrying nothing: GradSync must ignore it.
    announced = {i for kind, i in rec["events"] if kind == "pending"}
    delivers = lambda kind, i: kind == "ready" or (kind == "autograd" and i not in announced)   # noqa: E731
    n_contrib = {}
    for kind, i in rec["events"]:
        if delivers(kind, i):
            n_contrib[i] = n_contrib.get(i, 0) + 1
    assert max(n_contrib.values()) >= 3 and len(announced) >= 60 and len(n_contrib) > len(announced)
    for step in range(2):
        sync.begin_step()
        arena.grads.zero_()
        early = 0
        for kind, i in rec["events"]:
            p = arena.param_list[i]
            if kind == "pending":
                p._spn_grad_pending()
                continue
            if delivers(kind, i):
                # a contribution may never arrive after its bucket has left: that gradient would miss the all-reduce
                assert sync.bucket_of[i] not in sync.launched, (variant, kind, i, rec["names"][i])
                p.grad += float((rank + 1) * (1 + i % 7) + step)      # small integers: exact in bf16 as well
            if kind == "ready":
                p._spn_grad_ready()
            else:
                sync._autograd_ready(i)
            early = max(early, len(sync.launched))
        assert early >= len(sync.buckets) - 2, (early, len(sync.buckets))   # buckets leave during backward, not after it
        sync.finish()
        assert len(sync.launched) == len(sync.buckets)
        for i, p in enumerate(arena.param_list):
            want = sum(n_contrib.get(i, 0) * ((r + 1) * (1 + i % 7) + step) for r in range(world))
            assert torch.all(p.grad == want), (rec["names"][i], p.grad.flatten()[:3], want)
    q.put((rank, "ok"))
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["tiny", "tiny_xattn_mha"])
@pytest.mark.parametrize("grad_dtype", ["float32", "bfloat16"])
def test_gradsync_replays_the_real_models_event_order(variant, grad_dtype):
    """Two gloo ranks replay the recorded pending / ready order of the real model (tied tables, fused q|k|v, tied LM-head projection):
    no bucket leaves before its last contribution, buckets do leave during backward, every element is reduced exactly once -- with
    fp32 buckets and with the bf16 transport option."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_worker, args=(r, 2, port, variant, grad_dtype, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]
