"""Cost of one dependent kernel node under hipGraph replay on this machine: a chain of N one-thread kernels (spn_dec_add_pos) and a chain
of N small GEMVs, replayed; and the same chains launched on the stream without a graph.  Prints microseconds per node."""
import sys
import time

import torch

sys.path.insert(0, ".")
from scoreperformer_amd import ops  # noqa: E402


def replay_cost(body, n_nodes, reps=50):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            body()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_nodes * 1e6


def stream_cost(body, n_nodes, reps=20):
    body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_nodes * 1e6


def main():
    dev = torch.device("cuda")
    pos = torch.zeros(1, dtype=torch.int32, device=dev)
    N = 200

    def chain_tiny():
        for _ in range(N):
            ops.dec_add_pos(pos, 1)

    W = torch.randn(512, 512, device=dev)
    x = torch.randn(512, device=dev)
    y = torch.zeros(512, device=dev)

    def chain_gemv():
        for _ in range(N // 2):
            ops.dec_fused_gemv(W, x, y)
            ops.dec_fused_gemv(W, y, x)

    Wb = torch.randn(4096, 512, device=dev)
    yb = torch.zeros(4096, device=dev)

    def chain_gemv_big():
        for _ in range(N):
            ops.dec_fused_gemv(Wb, x, yb)

    print(f"one-thread kernel      : graph {replay_cost(chain_tiny, N):.2f} us/node, stream {stream_cost(chain_tiny, N):.2f} us/launch")
    print(f"GEMV 512x512 (128 blks): graph {replay_cost(chain_gemv, N):.2f} us/node, stream {stream_cost(chain_gemv, N):.2f} us/launch")
    print(f"GEMV 4096x512 (1024 b) : graph {replay_cost(chain_gemv_big, N):.2f} us/node, stream {stream_cost(chain_gemv_big, N):.2f} us/launch")


if __name__ == "__main__":
    main()
