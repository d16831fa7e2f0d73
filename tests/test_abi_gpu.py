"""GPU: the C ABI keeps what include/spn.h promises -- capturable (no allocation / synchronisation inside the library, caller-owned
workspaces) and re-entrant across host threads."""
import os
import subprocess
import sys
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CAPTURE_SCRIPT = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from scoreperformer_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
T, D, H = 4096, 512, 8
x = torch.randn(T, D, device=dev, generator=g)
w = (torch.randn(640, D, device=dev, generator=g) * 0.05).bfloat16()
dy = torch.randn(T, 640, device=dev, generator=g).bfloat16()
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
slopes = torch.tensor([2.0 ** -(i + 1) for i in range(H)], device=dev)
dw = torch.zeros(640, D, device=dev)

def step():
    # LayerNorm -> fused QKV projection -> attention fwd/bwd (ALiBi band buffer, multi-query) -> split-K weight gradient
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    qkv = ops.gemm(y, w).view(2, T // 2, 640)
    q, k, v = qkv[..., :512].unflatten(-1, (H, 64)), qkv[..., 512:576].unflatten(-1, (1, 64)), qkv[..., 576:].unflatten(-1, (1, 64))
    band = ops.attn_band_buffer(q, k)
    o, lse = ops.attn_fwd(q, k, v, slopes=slopes, causal=True, band=band)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :512].unflatten(-1, (H, 64)), dqkv[..., 512:576].unflatten(-1, (1, 64)), dqkv[..., 576:].unflatten(-1, (1, 64))
    ops.attn_bwd(q, k, v, o, torch.ones_like(o), lse, dq=dq, dk=dk, dv=dv, slopes=slopes, causal=True, band=band)
    ops.gemm(dy, y, ta=True, tb=True, out=dw.zero_())          # [640, 512] = dy^T y over K = 4096 tokens: split over K
    return o.float().sum() + dqkv.float().abs().sum() + dw.abs().sum()

# FIRST launches of the process happen under capture: any lazy allocation, attribute set-up or sync in the library would break it
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        captured = step()
torch.cuda.synchronize()
graph.replay(); torch.cuda.synchronize()
first = float(captured)
graph.replay(); torch.cuda.synchronize()
eager = float(step())
torch.cuda.synchronize()
assert first == float(captured), (first, float(captured))
assert abs(eager - first) <= 1e-5 * abs(eager), (eager, first)
print("CAPTURE_OK", first)
'''


def test_first_calls_of_a_process_are_capturable():
    """A fresh process captures LayerNorm + GEMM + attention fwd/bwd + a split-K weight gradient in a HIP graph before any eager call:
    the library allocates nothing and synchronises nothing (SURVEY.md §8(b) ownership / threading contract)."""
    r = subprocess.run([sys.executable, "-c", CAPTURE_SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_split_k_needs_no_library_memory(dev):
    """The weight-gradient GEMM gives the same answer with the caller's workspace and (unsplit) without one."""
    from scoreperformer_amd import ops
    from scoreperformer_amd.lib import call, ptr, stream_ptr, c_int, c_long, c_float
    import ctypes
    g = torch.Generator(device=dev).manual_seed(0)
    T, M, N = 32768, 512, 640
    a = torch.randn(T, M, device=dev, generator=g).bfloat16()
    b = torch.randn(T, N, device=dev, generator=g).bfloat16()
    want = a.float().t() @ b.float()
    split = ops.gemm(a, b, ta=True, tb=True, out_dtype=torch.float32)
    assert ops._gemm_ws_bytes(M, N, T, 1 | 2 | 4) > 0
    unsplit = torch.empty(M, N, device=dev)
    call("spn_gemm_bf16", ptr(a), ptr(b), ptr(unsplit), None, None, None, c_int(M), c_int(N), c_int(T), c_int(M), c_int(N), c_int(N),
         c_int(0), c_float(1.0), c_int(1 | 2 | 4), c_int(1), c_long(0), c_long(0), c_long(0), None, ctypes.c_size_t(0), stream_ptr())
    scale = want.abs().max()
    assert (split - want).abs().max() <= 2e-3 * scale and (unsplit - want).abs().max() <= 2e-3 * scale


def test_two_host_threads_call_the_library_concurrently(dev):
    """Two host threads drive GEMM (incl. split-K) and attention on their own streams at the same time; both match the serial results."""
    from scoreperformer_amd import ops
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(16384, 512, device=dev, generator=g).bfloat16()
    w = (torch.randn(1024, 512, device=dev, generator=g) * 0.05).bfloat16()
    dy = torch.randn(16384, 1024, device=dev, generator=g).bfloat16()
    q = torch.randn(2, 1024, 8, 64, device=dev, generator=g).bfloat16()
    kv = torch.randn(2, 1024, 1, 64, device=dev, generator=g).bfloat16()
    slopes = torch.tensor([2.0 ** -(i + 1) for i in range(8)], device=dev)

    def work():
        y = ops.gemm(a, w)
        dw = ops.gemm(dy, a, ta=True, tb=True, out_dtype=torch.float32)
        o, lse = ops.attn_fwd(q, kv, kv, slopes=slopes, causal=True)
        return y, dw, o, lse

    serial = work()
    torch.cuda.synchronize()
    results, errors = {}, []

    def runner(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                out = None
                for _ in range(8):
                    out = work()
            s.synchronize()
            results[i] = out
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for got, want in zip(results[i], serial):
            assert torch.equal(got, want)
