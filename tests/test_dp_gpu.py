"""GPU: the bucketed RCCL all-reduce path (GradSync) on a one-rank nccl group gives the same step as no sync at all."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gradsync_nccl_single_rank_matches_plain_step(dev):
    import torch.distributed as dist
    from oracle.weights import filled_state_dict
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.parallel import GradSync
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        batch = synthetic_batch(2, 64, seed=3, ragged=True, device=dev)
        results = []
        for use_sync in (False, True, "bf16", "spn"):   # "spn": libspn.so's own RCCL wrapper (spn_comm_allreduce) instead of dist.all_reduce
            model = ScorePerformer.init(model_config("tiny", dropout=0.0))
            model.load_state_dict(filled_state_dict(model, seed=5))
            arena = ParamArena(model, dev)
            model.train()
            opt = FusedAdamW(arena, lr=1e-3, weight_decay=0.0, grad_clip=1.0)
            sync = GradSync(arena, dist.group.WORLD, bucket_mb=0.05, force=True, transport="spn" if use_sync == "spn" else "torch",
                            grad_dtype=torch.bfloat16 if use_sync == "bf16" else torch.float32) if use_sync else None
            torch.manual_seed(11)
            first = None
            for it in range(2):
                if sync:
                    sync.begin_step()
                arena.zero_grad()
                out = model(**batch)
                out.loss.backward()
                if sync:
                    sync.finish()
                    assert len(sync.buckets) >= 3 and len(sync.launched) == len(sync.buckets)
                grads = arena.grads.clone()
                first = grads if it == 0 else first
                opt.step()
            torch.cuda.synchronize()
            results.append((float(out.loss.detach()), first, grads, arena.params.clone()))
        (l0, f0, g0, p0), (l1, f1, g1, p1), (l2, f2, g2, p2), (l3, f3, g3, p3) = results
        # native transport on one rank: the same buckets through ncclAllReduce on libspn.so's communication stream
        assert (f3 - f1).abs().max() <= 2e-5 * f1.abs().max() and (g3 - g1).abs().max() <= 5e-3 * g1.abs().max()
        # bf16 transport (one rank: the all-reduce is the identity): the arena holds the bf16-rounded fp32 gradients
        assert (f2 - f1).abs().max() <= 2.0 ** -8 * f1.abs().max()
        assert bool(((f2 - f1).abs() <= 2.0 ** -8 * f1.abs() + 1e-30).all())
        assert float((f2 != f1).float().mean()) > 0.5   # (it really went through bf16)
        # split-K weight gradients add with fp32 atomics: equal up to summation order on the first step, and up to that noise
        # carried through one optimizer step on the second
        assert (f0 - f1).abs().max() <= 2e-5 * f0.abs().max()
        assert abs(l0 - l1) < 3e-3      # second-step loss: Adam (lr 1e-3) amplifies the summation-order noise of near-zero gradients
        # (Adam turns tiny gradient differences of near-zero entries into O(lr) parameter differences: compare gradients only)
        assert (g0 - g1).abs().max() <= 5e-3 * g0.abs().max()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_the_global_batch_step(dev):
    """SURVEY.md §4 item 4 on one GPU: what two data-parallel ranks compute -- each rank the mean-loss gradient of ITS half of the batch,
    the all-reduce their SUM, the optimizer a 1/world scale folded into its kernel -- equals the single-GPU step on the whole batch.
    Holds exactly in exact arithmetic when every rank sees the same number of valid labels per key (full-length windows: SURVEY.md
    §8(e)) and the loss is a mean over samples; the MMD term is a per-rank estimate by design (kernel means are not additive), so its
    weight is 0 here.  The rank halves run one after the other on this GPU; their gradient arenas are summed as the all-reduce would."""
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    world = 2
    batch = synthetic_batch(4, 96, seed=17, device=dev)                       # full-length windows: equal label counts per half
    halves = [{k: v[r * 2:(r + 1) * 2] for k, v in batch.items()} for r in range(world)]

    def fresh():
        torch.manual_seed(5)
        cfg = model_config("tiny", dropout=0.0)
        cfg["perf_encoder"]["loss_weight"] = 0.0
        model = ScorePerformer.init(cfg)
        arena = ParamArena(model, dev)
        model.train()
        return model, arena, FusedAdamW(arena, lr=1e-3, weight_decay=1e-2, grad_clip=1.0)

    model, arena, opt = fresh()
    arena.zero_grad()
    model(**batch).loss.backward()
    g_global = arena.grads.clone()
    opt.step()
    p_global = arena.params.clone()

    model, arena, opt = fresh()
    total = torch.zeros_like(arena.grads)
    for half in halves:                      # rank r's backward ...
        arena.zero_grad()
        model(**half).loss.backward()
        total += arena.grads                 # ... and the all-reduce(SUM) of the gradient arenas
    arena.grads.copy_(total)
    for p in arena.param_list:
        p._spn_touched = True
    opt.step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    g_dp = total / world
    scale = g_global.abs().max()
    assert (g_dp - g_global).abs().max() <= 2e-2 * scale, float((g_dp - g_global).abs().max() / scale)   # bf16 GEMMs, different row sets
    rel = (g_dp - g_global).norm() / g_global.norm()
    assert float(rel) <= 2e-2, float(rel)
    # one Adam step (lr 1e-3) moves every entry by ~lr * sign(g): entries whose gradient is not at the noise level agree closely
    dp = (arena.params - p_global).abs()
    assert float(dp.max()) <= 2.5e-3 and float((dp > 2e-4).float().mean()) < 0.05, (float(dp.max()), float((dp > 2e-4).float().mean()))


def test_two_rank_step_with_the_mmd_path_on_matches_the_two_half_oracle(dev):
    """C4 (hierarchical MMD-VAE path + data parallelism) as a combination: with the MMD weight ON every rank's loss contains ITS OWN
    kernel-mean estimates (its half of the batch against its own N(0, I) samples; mmd_transformer.py:505-534), and the data-parallel
    gradient is the SUM of the ranks' gradients.  The HIP path runs the two halves (one after the other on this GPU, arenas summed as the
    all-reduce sums them); the fp32 CPU oracle runs the same two halves with the same samples: per-rank MMD entries within 1e-3, per-rank
    losses within 1e-3, the summed gradient within 5 % relative L2 as one vector (bf16 GEMM operands)."""
    from oracle import ref_cpu
    from oracle.weights import canonical
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    world = 2
    cfg = model_config("tiny", dropout=0.0)
    assert float(cfg["perf_encoder"]["loss_weight"]) == 1.0
    torch.manual_seed(8)
    model = ScorePerformer.init(model_config("tiny", dropout=0.0))   # the model's own initialisation, as a training run starts (smoke())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    arena = ParamArena(model, dev)
    model.train()
    batch = synthetic_batch(2 * world, 64, seed=17, ragged=True)
    zs = [[torch.randn(256, d, generator=torch.Generator().manual_seed(1000 * r + i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
          for r in range(world)]
    total = torch.zeros_like(arena.grads)
    got = []
    for r in range(world):
        half = {k: v[r * 2:(r + 1) * 2].to(dev) for k, v in batch.items()}
        model.perf_encoder._z_override = [t.to(dev) for t in zs[r]]
        arena.zero_grad()
        out = model(**half)
        out.loss.backward()
        total += arena.grads
        got.append({k: float(v) for k, v in out.losses.items()} | {"loss": float(out.loss)})
    torch.cuda.synchronize()
    leaves, sdg = {}, {}
    for k, v in sd.items():
        leaf = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("token_values") else v
        sdg[k] = leaves.setdefault(canonical(k), leaf)
    for r in range(world):
        half = {k: v[r * 2:(r + 1) * 2] for k, v in batch.items()}
        ref = ref_cpu.score_performer_forward(sdg, cfg, half, zs[r], training=True)
        ref["loss"].backward()                      # accumulates over the two halves: the all-reduce's sum
        assert abs(got[r]["loss"] - float(ref["loss"].detach())) <= 1e-3
        mmd_keys = [k for k in ref["losses"] if k.startswith("MMD/")]
        assert len(mmd_keys) == 4
        for k in mmd_keys:
            assert abs(got[r][k] - float(ref["losses"][k].detach())) <= 1e-3, (r, k)
    assert any(abs(got[0][k] - got[1][k]) > 1e-6 for k in got[0] if k.startswith("MMD/"))   # per-rank estimates, not one shared value
    err2 = ref2 = 0.0
    seen = set()
    for (k, p), off in zip(zip(arena.names, arena.param_list), arena.offsets):
        g_ref = sdg[k].grad
        if g_ref is None or id(p) in seen:
            continue
        seen.add(id(p))
        g = total[off:off + p.numel()].view(p.shape).float().cpu()
        err2 += float((g - g_ref).double().pow(2).sum()); ref2 += float(g_ref.double().pow(2).sum())
    assert len(seen) > 100 and err2 ** 0.5 <= 0.05 * ref2 ** 0.5, (err2 ** 0.5, ref2 ** 0.5)


def test_native_comm_allreduce_is_stream_ordered(dev):
    """spn_comm_*: one-rank communicator bound to the RCCL copy PyTorch loaded; the all-reduce (identity on one rank, fp32 and bf16)
    runs on the library's own stream, ordered behind the producer kernel and ahead of the consumer through events only."""
    from scoreperformer_amd.comm import NativeComm, unique_id
    comm = NativeComm(1, 0, unique_id())
    try:
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.zeros(1 << 22, device=dev, dtype=dtype)
            x.add_(3.0)                     # producer on the current stream
            comm.all_reduce_(x)             # must see the 3.0
            comm.wait()
            y = x * 2.0                     # consumer on the current stream
            torch.cuda.synchronize()
            assert float(y.float().min()) == 6.0 and float(y.float().max()) == 6.0
    finally:
        comm.close()


def _two_process_worker(rank, world, port, q):
    """One of two REAL processes sharing the box's GPU: the bench's own step (GradSync.begin_step / forward / backward with the bucketed
    all-reduce launched from inside backward / finish / FusedAdamW with grad_scale = 1 / world) over a gloo group -- gloo reduces CUDA
    tensors through the host, so two ranks can run on one device, which RCCL refuses."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.weights import filled_state_dict
        from scoreperformer_amd.arena import ParamArena, FusedAdamW
        from scoreperformer_amd.models import ScorePerformer
        from scoreperformer_amd.parallel import GradSync
        from scoreperformer_amd.synthetic import model_config, synthetic_batch

        def fresh():
            model = ScorePerformer.init(model_config("tiny", dropout=0.0))
            model.load_state_dict(filled_state_dict(model, seed=5))
            arena = ParamArena(model, dev)
            model.train()
            return model, arena

        batch = synthetic_batch(2, 64, seed=40 + rank, ragged=True, device=dev, with_bounds=True)   # every rank its own batch
        z = [torch.randn(256, d, generator=torch.Generator().manual_seed(7 * rank + i)).to(dev) for i, d in enumerate((32, 20, 8, 4))]
        # this rank's own gradient, no synchronisation
        model, arena = fresh()
        model.perf_encoder._z_override = z
        model(**batch).loss.backward()
        local = arena.grads.clone()
        # the data-parallel step
        model, arena = fresh()
        model.perf_encoder._z_override = z
        opt = FusedAdamW(arena, lr=1e-3, weight_decay=0.0, grad_clip=1.0)
        sync = GradSync(arena, dist.group.WORLD, bucket_mb=0.05, transport="torch")
        assert sync.active and sync.world == 2 and len(sync.buckets) >= 3
        sync.begin_step()
        out = model(**batch)
        out.loss.backward()
        during = len(sync.launched)                     # buckets that left while backward was still running
        sync.finish()
        assert len(sync.launched) == len(sync.buckets)
        reduced = arena.grads.clone()
        active = arena.active_params()
        assert all(a or not p_.requires_grad for a, p_ in zip(active, arena.param_list))   # replicas step the same parameter set
        opt.step(grad_scale=1.0 / world)
        for _ in range(4):                              # four more steps: the replicas must not drift (fixed-order gradient norm)
            sync.begin_step()
            model(**batch).loss.backward()
            sync.finish()
            opt.step(grad_scale=1.0 / world)
        torch.cuda.synchronize()
        # both ranks' local gradients, to rank 0's check
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = gathered[0] + gathered[1]
        err = float((reduced - want).abs().max() / want.abs().max())
        params = arena.params.clone()
        plist = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(plist, params)
        rl = [torch.empty_like(reduced) for _ in range(world)]
        dist.all_gather(rl, reduced)
        diff = (plist[0] - plist[1]).abs()
        bad = [(arena.names[i], float(diff[o:o + p_.numel()].max())) for i, (p_, o) in enumerate(zip(arena.param_list, arena.offsets))
               if float(diff[o:o + p_.numel()].max()) > 0]
        print(f"rank {rank}: reduced grads equal across ranks: {bool(torch.equal(rl[0], rl[1]))}; inactive here: "
              f"{[arena.names[i] for i, a in enumerate(active) if not a]}; differing parameters: {bad[:8]}", flush=True)
        q.put((rank, err, during, len(sync.buckets), bool(torch.equal(plist[0], plist[1])), float(out.loss.detach())))
        sync.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu_run_the_data_parallel_step_over_gloo():
    """The first data-parallel run with MORE THAN ONE RANK and the real kernels: two processes on this box's GPU (gloo carries the CUDA
    buckets; RCCL cannot put two ranks on one device).  Every rank's arena ends with the SUM of the two local gradients (2e-5: split-K
    and scatter atomics reorder sums between the two runs of a rank), buckets leave while backward is still running, and after the
    optimizer step (1 / world folded into the kernel) both ranks hold bit-identical parameters."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_process_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        if p.exitcode is None:
            p.kill()
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, err, during, nb, same, loss in res:
        assert err <= 2e-5, (rank, err)
        assert 0 < during <= nb, (rank, during, nb)      # some buckets were launched from inside backward
        assert same, "ranks diverged after the optimizer step"
        assert loss == loss
    assert res[0][5] != res[1][5]                        # (the two ranks really saw different batches)


def test_bench_runs_with_two_ranks_on_one_gpu_over_gloo():
    """bench.py's own N > 1 path end to end -- launcher environment, process group, replicas from one seed, per-rank batches, the bucketed
    all-reduce from inside backward, max-over-ranks timing, the instrumented roofline steps on every rank, rank 0's ONE JSON line, an
    orderly shutdown -- with two ranks on this box's GPU (`--dist-backend gloo --one-device`: test aids; RCCL itself needs a device per
    rank and is first exercised by the driver's scaling run).  Tiny model, short sequences: the numbers mean nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--standalone", "--local-addr", "127.0.0.1",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preset", "tiny", "--batch", "4", "--seq", "256",
           "--dist-backend", "gloo", "--one-device"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 alone prints, and prints one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert d["config"]["dp_transport"] == "torch" and d["config"]["gemm_persist_bwd"] == 0     # no persistent backward walk under an all-reduce
    assert d["value"] > 0 and d["ms_per_step"] > 0 and abs(d["value"] - 2 * 4 * 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "roofline" in d and "cpu_baseline" not in d and "dp1_forced" not in d              # N = 1 objects stay out of an N > 1 line
    assert d["config"]["final_loss"] == d["config"]["final_loss"]
