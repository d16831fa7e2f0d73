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
