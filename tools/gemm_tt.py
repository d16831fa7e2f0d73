"""A/B of split-K output strategies on the weight-gradient shapes.  env LIB = library under tools/_bin."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "tools/_bin", os.environ.get("LIB", "libspn_base.so")))
dev = torch.device("cuda"); P = ctypes.c_void_p; T = 131072
for M, N in [(4096, 512), (512, 2048), (640, 512), (512, 512), (1536, 512)]:
    a = torch.randn(T, M, device=dev).bfloat16(); b = torch.randn(T, N, device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev)
    def run():
        return lib.spn_gemm_bf16(P(a.data_ptr()), P(b.data_ptr()), P(c.data_ptr()), None, None, None, M, N, T, M, N, N, 0,
                                 ctypes.c_float(1.0), 1 | 2 | 4 | 8, 1, ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0), None, ctypes.c_size_t(0), None)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{os.environ.get('LIB','')} dW {M}x{N}: {ms:.3f} ms {2.0*M*N*T/ms/1e9:.0f} TF/s")
