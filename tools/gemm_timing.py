"""Tuning aid: per-segment s_memtime sums of the ping-pong GEMM (needs libspn_timing.so built with -DSPN_GEMM_TIMING)."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "tools/_bin", os.environ.get("LIB", "libspn_timing.so")))
E = lambda k, d: int(os.environ.get(k, d))
M, N, K = E("M", 8192), E("N", 8192), E("K", 8192)
dev = torch.device("cuda")
FLAGS = E("FLAGS", 0)   # bit0: A stored [K][M], bit1: B stored [K][N], bit2: fp32 C
TA, TB, F32 = FLAGS & 1, FLAGS & 2, FLAGS & 4
LDA, LDB = E("LDA", M if TA else K), E("LDB", N if TB else K)
a = torch.randn((K, M) if TA else (M, K), device=dev).bfloat16(); b = torch.randn((K, N) if TB else (N, K), device=dev).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.float32 if F32 else torch.bfloat16)
dbg = torch.zeros(32, device=dev, dtype=torch.int64)
if os.environ.get("OW"):   # one-wave-per-SIMD kernels: segments are block 0 / wait + barrier / block 1 / loop overhead per K tile of 32
    pass   # (round 6: the one-wave kernels are a variant build -- SPN_LIB=tools/_bin/<ow variant>.so -- not a knob)
if hasattr(lib, "spn_gemm_set_debug"):
    lib.spn_gemm_set_debug(ctypes.c_void_p(dbg.data_ptr()))
P = ctypes.c_void_p
for _ in range(2):
    rc = lib.spn_gemm_bf16(P(a.data_ptr()), P(b.data_ptr()), P(c.data_ptr()), None, None, None, M, N, K, LDA, LDB, N, 0,
                           ctypes.c_float(1.0), FLAGS, 1, ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0), None, ctypes.c_size_t(0), None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.spn_gemm_bf16(P(a.data_ptr()), P(b.data_ptr()), P(c.data_ptr()), None, None, None, M, N, K, LDA, LDB, N, 0,
                      ctypes.c_float(1.0), FLAGS, 1, ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0), None, ctypes.c_size_t(0), None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
if FLAGS == 0 and LDA == K and LDB == K:
    ref = a[:512].float() @ b.float().t()
    err = ((c[:512].float() - ref).abs().max() / ref.abs().max()).item()
    ref2 = a[-256:].float() @ b[-256:].float().t()
    err2 = ((c[-256:, -256:].float() - ref2).abs().max() / ref2.abs().max()).item()
    print("relerr", f"{err:.2e} {err2:.2e}", "OK" if max(err, err2) < 1e-2 else "WRONG")
print(f"{os.environ.get('LIB', '')} M={M} N={N} K={K}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.0f} TF/s")
names = ["ds_read issue", "dma issue", "vmcnt wait", "barrier A", "lgkm wait", "mfma", "barrier B", "-"]
if os.environ.get("OW"):
    d = dbg.cpu().tolist(); nt = K // 32
    for w, off in ((0, 0), (3, 8)):
        print(f"wave {w}: per K tile of 32: block 0 {d[off]/nt:.0f}  lgkmcnt(0)+barrier {d[off+1]/nt:.0f}  block 1 {d[off+2]/nt:.0f}  between tiles {d[off+3]/nt:.0f}   (32 MFMAs = 1024 cycles of matrix pipe)")
    print(f"main loop total {d[16]} cycles for {nt} K tiles = {d[16]/nt:.0f} per tile")
    sys.exit(0)
d = dbg.cpu().tolist(); ph = 4 * K // 64
print(f"block (0,0): prologue + main loop {d[16]} cycles, epilogue (stores acknowledged) {d[17]} cycles")
for g in range(2):
    print(f"group {g}: " + "  ".join(f"{names[k]}={d[g*8+k]/ph:.0f}" for k in range(7)), " total/phase", sum(d[g*8:g*8+7]) / ph)
