"""Micro-benchmark of the segment kernels of the hierarchical heads at C3 (b 64, n 2048, d 512): one pass for all levels against one
launch per level.   python tools/bench_seg.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd import ops
from scoreperformer_amd.synthetic import synthetic_batch

dev = torch.device("cuda")
b, n, d = 64, 2048, 512
batch = synthetic_batch(b, n, seed=1, device=dev)
mask = batch["perf_mask"]
segs = [(~mask).long(), batch["bars"], batch["beats"], batch["onsets"]]
S = [2] + [int(s.max()) + 1 for s in segs[1:]]
counts = [ops.segment_count(s, k) for s, k in zip(segs, S)]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(b, n, d, device=dev).to(dt)
    outs = [torch.zeros(b, k, d, device=dev) for k in S]
    per = [timed(lambda i=i: ops.segment_sum(x, segs[i], S[i], counts=counts[i], rowmask=mask, out=outs[i])) for i in range(4)]
    multi = timed(lambda: ops.segment_sum_multi(x, mask, segs, counts, outs, S))
    singles = [timed(lambda i=i: ops.segment_sum_multi(x, mask, [segs[i]], [counts[i]], [outs[i]], [S[i]])) for i in range(4)]
    print(f"{str(dt):16s} per level {['%.0f' % v for v in per]} us (sum {sum(per):.0f})   multi(4) {multi:.0f} us   multi(1 level each) {['%.0f' % v for v in singles]}")
srcs = [torch.randn(b, k, d, device=dev) for k in S]
y = torch.zeros(b, n, d, device=dev)
per = [timed(lambda i=i: ops.segment_gather(srcs[i], segs[i], counts=counts[i], out=y, accumulate=True)) for i in range(4)]
multi = timed(lambda: ops.segment_gather_multi(srcs, segs, counts, S, mask, d))
print(f"gather: per level {['%.0f' % v for v in per]} us (sum {sum(per):.0f})   multi {multi:.0f} us")
