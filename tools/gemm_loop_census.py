#!/usr/bin/env python3
"""Instruction census of the steady-state K loop of gemm_pp_kernel (one K tile of 64 per iteration and wave) from a hipcc -S dump.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude scoreperformer_amd/csrc/gemm.hip -o /tmp/gemm.s
    python tools/gemm_loop_census.py /tmp/gemm.s

For every instantiation: the basic block that holds exactly 32 MFMAs and ends in the loop's back edge, by instruction class."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN\w*gemm_pp_kernel\w*:", l)]
for st in starts:
    name = lines[st].split(":")[0]
    end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[st:end]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur); cur = (t.split(":")[0], [])
        elif t and not t.startswith((";", ".")) and not t.endswith(":"):
            cur[1].append(t.split(";")[0].strip())
    blocks.append(cur)
    for label, ins in blocks:
        nm = sum(i.startswith("v_mfma") for i in ins)
        if nm not in (32, 64) or not any(i.startswith("s_cbranch") and label in i for i in ins):
            continue
        c = collections.Counter()
        for i in ins:
            op = i.split()[0]
            c["mfma" if op.startswith("v_mfma") else "lds_read" if op.startswith("ds_read") else "lds_dma" if op.startswith("buffer_load") else
              "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op == "s_barrier" else "valu" if op.startswith("v_") else
              "s_nop" if op == "s_nop" else "salu" if op.startswith("s_") else "other"] += 1
        reads = collections.Counter(i.split()[0] for i in ins if i.startswith("ds_read"))
        print(f"{name}\n   loop {label} ({nm // 32} K tile(s) per iteration): {len(ins)} instructions  {dict(c)}  reads {dict(reads)}")
