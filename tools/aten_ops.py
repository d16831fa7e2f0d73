"""Which torch (ATen) kernels still run inside one C3 train step, with input shapes and the Python frame that launched them.

    python tools/aten_ops.py [--batch 64 --seq 2048]

The libspn kernels are launched through ctypes and never show up as aten ops; what is listed here is glue (adds, fills, copies) that a
fused kernel or an arena view could absorb.
"""
import argparse
import collections
import sys

import torch

sys.path.insert(0, ".")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--dropout", type=float, default=0.1)
    args = ap.parse_args()
    from scoreperformer_amd.arena import ParamArena, FusedAdamW
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.parallel import GradSync
    from scoreperformer_amd.synthetic import model_config, synthetic_batch

    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    cfg = model_config("c3", max_seq_len=max(args.seq, 256), dropout=args.dropout, latent_dropout=[0.0, 0.1, 0.2, 0.4])
    model = ScorePerformer.init(cfg)
    arena = ParamArena(model, dev)
    model.train()
    model.sync_free = True
    opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-6, grad_clip=2.0)
    sync = GradSync(arena, None, persistent_backward=True)
    batch = synthetic_batch(args.batch, args.seq, seed=1234, device=dev)
    model.perf_encoder.segment_bounds = {m: int(batch[k].max()) + 1 for m, k in
                                         (("bar_mean", "bars"), ("beat_mean", "beats"), ("onset_mean", "onsets"))}

    def step():
        sync.begin_step()
        out = model(**batch)
        out.loss.backward()
        sync.finish()
        opt.step(grad_scale=1.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    # call sites: a dispatch mode on the calling thread (the backward is kept on it too) logs every aten op that touches the GPU
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.name()
            if any(s in name for s in ("view", "reshape", "as_strided", "detach", "alias", "slice", "select", "transpose", "expand", "permute",
                                       "squeeze", "unbind", "split", "t.default", "empty", "_local_scalar", "is_", "size", "stride")):
                return out
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            if not any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + ([out] if isinstance(out, torch.Tensor) else [])):
                return out
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "scoreperformer_amd" in fr.filename:
                    where = f"{fr.filename.split('scoreperformer_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            sites[(name, str(shapes)[:60], where)] += 1
            return out

    with torch.autograd.set_multithreading_enabled(False), Log():
        step()
    torch.cuda.synchronize()
    print("== aten ops by call site (one step)")
    for (name, shapes, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
        print(f"n={n:4d}  {name:32s} {shapes:60s} {where}")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
            continue
        if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
            continue   # count the leaf op only
        key = (ev.name, str(ev.input_shapes)[:70], "")
        rows[key][0] += 1
        rows[key][1] += ev.device_time_total
    total = sum(v[1] for v in rows.values())
    print(f"aten kernels in one step: {sum(v[0] for v in rows.values())} launches, {total / 1e3:.2f} ms of GPU time")
    for key, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{us / 1e3:7.3f} ms  n={n:4d}  {key[0]:28s} {key[1]:70s} {key[2]}")


if __name__ == "__main__":
    main()
