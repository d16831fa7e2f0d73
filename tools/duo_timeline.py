"""Tuning aid: when do the workgroups of the gated-backward GEMM (two 4-wave workgroups per CU) run their main loops and their
epilogues?  Builds csrc/gemm.hip with -DSPN_GEMM_TIMING into tools/_bin/libspn_timing.so (every workgroup stamps start / end of
main loop / end of epilogue with the 100 MHz real-time counter plus its HW_ID), runs spn_gemm_glu_bwd at the benchmark's FFN shape
and prints, per phase, durations and how much of a workgroup's epilogue ran while the OTHER workgroup of its CU was in its main loop."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "scoreperformer_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_bin", "timing_spn.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "_bin", "gemm_timing.o")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DSPN_GEMM_TIMING", "-I", os.path.join(ROOT, "include")]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, "gemm.hip"), "-o", obj])
    others = [os.path.join(CSRC, "_obj", f) for f in os.listdir(os.path.join(CSRC, "_obj")) if f.endswith(".o") and f != "gemm.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, obj] + others)


def main():
    if "--build" in sys.argv:
        build()
        return
    lib = ctypes.CDLL(OUT)
    duo = float(os.environ.get("DUO", 2))     # 2: two 8-wave workgroups per CU (default), 1: two 4-wave workgroups (with per-block epilogue stamps)
    lib.spn_set_tuning(b"glu_bwd_duo", ctypes.c_double(duo))
    print(f"glu_bwd_duo = {duo:.0f}")
    M, I, K = 131072, 2048, 512
    p_drop = float(os.environ.get("P", 0.1))
    dev = torch.device("cuda")
    dy = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w2 = (torch.randn(K, I, device=dev) * K ** -0.5).bfloat16()
    u = torch.randn(M, 2 * I, device=dev).bfloat16()
    du = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16)
    part = torch.zeros((M + 127) // 128, 2 * I, device=dev)
    nwg = (I // 128) * (M // 256)
    dbg = torch.zeros(32 + 18 * nwg, device=dev, dtype=torch.int64)
    P = ctypes.c_void_p

    def run():
        rc = lib.spn_gemm_glu_bwd(P(dy.data_ptr()), P(w2.data_ptr()), P(u.data_ptr()), P(du.data_ptr()), P(part.data_ptr()), M, I, K, K, I, 2 * I,
                                  2 * I, 0, ctypes.c_float(p_drop), ctypes.c_uint(5), P(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    lib.spn_gemm_set_debug(P(dbg.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    t = dbg[32:32 + 5 * nwg].view(nwg, 5).cpu()
    est = dbg[32 + 5 * nwg:].view(nwg, 13).cpu()
    t0 = int(t[:, 0].min())
    start, loop, end = (t[:, 0] - t0).double() / 100.0, (t[:, 1] - t0).double() / 100.0, (t[:, 2] - t0).double() / 100.0   # microseconds
    hw, xcc = t[:, 3], t[:, 4] & 0xf
    cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf)     # (xcc, se, sh, cu)
    print(f"launch {e0.elapsed_time(e1) * 1e3:.0f} us by events, {float(end.max()):.0f} us by stamps; {nwg} workgroups on {len(set(cu.tolist()))} CUs")
    main_us, epi_us = loop - start, end - loop
    q = lambda x: [round(float(v), 1) for v in torch.quantile(x, torch.tensor([0.05, 0.5, 0.95], dtype=torch.float64))]
    print("main loop us (5/50/95 %):", q(main_us), " epilogue us:", q(epi_us), " tile us:", q(end - start))
    rel = (est[:, 1:] - est[:, :-1]).double() / 100.0
    names = ["b0 wait u", "b0 math", "b0 stage+store", "b1 wait u", "b1 math", "b1 stage+store", "b2 wait u", "b2 math", "b2 stage+store",
             "b3 wait u", "b3 math", "b3 stage+store"]
    print("epilogue of wave 0, median us per segment:", {n: round(float(rel[:, k].median()), 2) for k, n in enumerate(names)},
          "sum", round(float(rel.sum(1).median()), 1))
    # per CU: overlap of each workgroup's epilogue with main loops of other workgroups on the same CU
    by_cu = {}
    for i, c in enumerate(cu.tolist()):
        by_cu.setdefault(c, []).append(i)
    tot_epi = tot_ov = tot_both_main = tot_main = 0.0
    busy = []
    for c, ids in by_cu.items():
        iv_main = [(float(start[i]), float(loop[i])) for i in ids]
        iv_epi = [(float(loop[i]), float(end[i])) for i in ids]
        for k, (a, b) in enumerate(iv_epi):
            tot_epi += b - a
            for k2, (c0, c1) in enumerate(iv_main):
                if k2 != k:
                    tot_ov += max(0.0, min(b, c1) - max(a, c0))
        for k, (a, b) in enumerate(iv_main):
            tot_main += b - a
            for k2, (c0, c1) in enumerate(iv_main):
                if k2 != k:
                    tot_both_main += max(0.0, min(b, c1) - max(a, c0))
        busy.append(len(ids))
    print(f"epilogue time overlapped by another workgroup's main loop on the same CU: {tot_ov / tot_epi:.2f}; "
          f"main-loop time shared with another main loop: {tot_both_main / tot_main:.2f}; workgroups per CU {min(busy)}..{max(busy)}")
    # chip-wide: how many workgroups are in their epilogue at a time (sampled every us)
    T = int(float(end.max())) + 1
    grid = torch.arange(T, dtype=torch.float64)[:, None]
    n_epi = ((grid >= loop[None, :]) & (grid < end[None, :])).sum(1)
    n_main = ((grid >= start[None, :]) & (grid < loop[None, :])).sum(1)
    print("workgroups in epilogue, sampled every 25 us:", n_epi[::25].tolist())
    print("workgroups in main loop, sampled every 25 us:", n_main[::25].tolist())


if __name__ == "__main__":
    main()
