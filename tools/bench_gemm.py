"""A/B micro-benchmark of the bf16 GEMM kernels at the C3 shapes: ping-pong (256x256, one workgroup per CU) against duo (256x128, two
workgroups per CU), interleaved rounds in one process, HIP events; hipBLASLt (torch.matmul, bf16 out) as the yardstick.
Usage: python tools/bench_gemm.py [--rounds 5] [--reps 10] [--no-blas]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops, lib

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--no-blas", action="store_true")
ap.add_argument("--only", default="")
ap.add_argument("--variants", default="pp:gemm_duo=0;duo:gemm_duo=2",
                help="name:knob=value,knob=value;name:... (knobs of csrc/tuning.h; unspecified knobs keep their defaults)")
args = ap.parse_args()
VARIANTS = []
for item in args.variants.split(";"):
    name, _, kv = item.partition(":")
    VARIANTS.append((name, [(k, float(v)) for k, v in (x.split("=") for x in kv.split(",") if x)]))
ALL_KNOBS = sorted({k for _, kvs in VARIANTS for k, _ in kvs})
DEFAULTS = {k: lib.get_tuning(k) for k in ALL_KNOBS}


def select(kvs):
    for k in ALL_KNOBS:
        lib.set_tuning(k, DEFAULTS[k])
    for k, v in kvs:
        lib.set_tuning(k, v)

dev = torch.device("cuda")
T = 131072


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def check(out, a, b, ta, tb, bias, residual, tag):
    am, bm = (a.t() if ta else a), (b if tb else b.t())
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(out.shape[0] - 300, out.shape[0], device=dev)]) if out.shape[0] > 600 else torch.arange(out.shape[0], device=dev)
    ref = am[rows].float() @ bm.float()
    if bias is not None:
        ref = ref + bias
    if residual is not None:
        ref = ref + residual[rows]
    err = ((out[rows].float() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-2, (tag, err)
    return err


def run(M, N, K, ta, tb, f32, extras=False):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if extras else None
    residual = torch.randn(M, N, device=dev, generator=g) if (extras and f32) else None
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    fn = lambda: ops.gemm(a, b, ta=ta, tb=tb, out=out, bias=bias, residual=residual)
    res = {}
    errs = {}
    for name, kvs in VARIANTS:
        select(kvs)
        out.zero_(); fn(); torch.cuda.synchronize()
        errs[name] = check(out, a, b, ta, tb, bias, residual, name)
        res[name] = []
    for _ in range(args.rounds):
        for name, kvs in VARIANTS:
            select(kvs)
            fn(); res[name].append(timed(fn, args.reps))
    line = f"M={M:6d} N={N:5d} K={K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'} {'f32' if f32 else 'bf16'}{'+b+r' if extras else '    '}:"
    for name, _ in VARIANTS:
        ms = statistics.median(res[name])
        line += f"  {name} {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF/s (err {errs[name]:.1e})"
    if not args.no_blas:
        am, bm = (a.t() if ta else a), (b if tb else b.t())
        o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f2 = lambda: torch.matmul(am, bm, out=o2)
        f2(); ms = timed(f2, args.reps)
        line += f"  | hipBLASLt bf16-out {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF/s"
    print(line, flush=True)


def run_glu(M, I, K, p_drop):
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(2 * I, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(2 * I, device=dev, generator=g)
    fn = lambda: ops.gemm_glu(x, w, bias, act=0, p_drop=p_drop, seed=11)
    outs, res = {}, {name: [] for name, _ in VARIANTS}
    for name, kvs in VARIANTS:
        select(kvs)
        outs[name] = fn(); torch.cuda.synchronize()
    first = VARIANTS[0][0]
    same = all(torch.equal(outs[first][0], outs[n][0]) and torch.equal(outs[first][1], outs[n][1]) for n, _ in VARIANTS)
    for _ in range(args.rounds):
        for name, kvs in VARIANTS:
            select(kvs)
            fn(); res[name].append(timed(fn, args.reps))
    line = f"M={M:6d} I={I:5d} K={K:6d} GLU p={p_drop}:"
    for name, _ in VARIANTS:
        ms = statistics.median(res[name])
        line += f"  {name} {ms*1e3:7.1f} us {2.0*M*2*I*K/ms/1e9:6.0f} TF/s"
    print(line + f"  bit-identical u,g: {same}", flush=True)


shapes = [
    ("w1", (T, 4096, 512, False, False, False, False)),
    ("w2", (T, 512, 2048, False, False, True, True)),
    ("qkv", (T, 640, 512, False, False, False, False)),
    ("out", (T, 512, 512, False, False, True, True)),
    ("w2d", (T, 2048, 512, False, True, False, False)),
    ("w1d", (T, 512, 4096, False, True, False, False)),
    ("qkvd", (T, 512, 640, False, True, False, False)),
    ("emb", (T, 512, 1536, False, False, False, True)),
    ("ada", (T, 1024, 64, False, False, True, True)),
    ("adad", (T, 64, 1024, False, True, False, False)),
    ("big", (8192, 8192, 8192, False, False, False, False)),
    ("w1g", (4096, 512, T, True, True, True, False)),      # weight gradients (split-K)
    ("w2g", (512, 2048, T, True, True, True, False)),
    ("qkvg", (640, 512, T, True, True, True, False)),
]
for tag, sh in shapes:
    if args.only and tag not in args.only.split(","):
        continue
    run(*sh)
if not args.only or "glu" in args.only.split(","):
    run_glu(T, 2048, 512, 0.0)
    run_glu(T, 2048, 512, 0.1)
select([])
