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


# ---- C4: the hierarchical MMD-VAE path under data parallelism -----------------------------------------------------------------------
# BASELINE config 4 (style-encoder VAE with the MMD loss, DP).  The MMD term is a kernel mean over the rank's own latents and its own
# N(0, I) samples (mmd_transformer.py:505-534): it is NOT additive over a batch split, so the data-parallel step is DEFINED as every rank's
# own loss (token losses + its per-rank MMD estimates) on its half, gradients summed by the all-reduce, 1/world in the optimizer
# (SURVEY.md §8(e)).  Two gloo ranks run the fp32 CPU oracle of the tiny model on their halves with the MMD weight ON, send the
# gradients through GradSync's buckets, and rank 0 checks the reduced arena against the two-half computation done in one process.

def _oracle_half(rank, world):
    from oracle import ref_cpu
    from oracle.weights import canonical, filled_state_dict
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    torch.set_num_threads(1)
    cfg = model_config("tiny", dropout=0.0)
    assert float(cfg["perf_encoder"]["loss_weight"]) == 1.0          # the MMD path is on
    sd = filled_state_dict(ScorePerformer.init(model_config("tiny", dropout=0.0)), seed=5)
    batch = synthetic_batch(2 * world, 64, seed=17)
    half = {k: v[rank * 2:(rank + 1) * 2] for k, v in batch.items()}
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(1000 * rank + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    leaves, sdg = {}, {}
    for k, v in sd.items():
        leaf = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v
        sdg[k] = leaves.setdefault(canonical(k), leaf)
    out = ref_cpu.score_performer_forward(sdg, cfg, half, z, training=True)
    out["loss"].backward()
    names = sorted(k for k, v in leaves.items() if v.requires_grad and v.grad is not None)
    return names, [leaves[k].grad.detach().clone() for k in names], {k: float(v.detach()) for k, v in out["losses"].items()}


def _c4_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scoreperformer_amd.parallel import GradSync
    names, grads, losses = _oracle_half(rank, world)
    mmd = {k: v for k, v in losses.items() if k.startswith("MMD/")}
    assert len(mmd) == 4 and all(v > 0 for v in mmd.values()), losses
    arena = FakeArena([g.numel() for g in grads])
    sync = GradSync(arena, dist.group.WORLD, bucket_mb=0.05)
    assert len(sync.buckets) >= 4
    sync.begin_step()
    for p in arena.param_list:
        p._spn_grad_pending()
    for p, g in reversed(list(zip(arena.param_list, grads))):
        p.grad += g.view(p.shape)
        p._spn_grad_ready()
    sync.finish()
    # every rank's MMD estimate is its own: gather them and make sure they differ (different halves, different samples)
    box = [None] * world
    dist.all_gather_object(box, mmd)
    assert box[0] != box[1]
    if rank == 0:
        other_names, other, other_losses = _oracle_half(1, world)
        assert other_names == names and {k: v for k, v in other_losses.items() if k.startswith("MMD/")} == box[1]
        for p, g0, g1, k in zip(arena.param_list, grads, other, names):
            want = (g0 + g1).view(p.shape)
            assert torch.equal(p.grad, want), (k, float((p.grad - want).abs().max()))
        # the MMD gradient is really in there: the latent heads get gradient from nothing else but MMD + the decoder's style path
        assert any("vae_head" in k or "latent" in k for k in names)
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_c4_mmd_path_under_two_rank_data_parallelism():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]
