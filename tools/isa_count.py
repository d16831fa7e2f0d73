#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S dump (tuning aid).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude scoreperformer_amd/csrc/gemm.hip -o /tmp/gemm.s
    python tools/isa_count.py /tmp/gemm.s 'gemm_duo_kernelILb1EtLi3E' [--after-last-mfma]

Prints per class: vector ALU, transcendental, scalar, LDS, vector memory, MFMA, s_nop, branches, waits; with --after-last-mfma only
the part behind the last v_mfma (the tile epilogue of the GEMM kernels)."""
import collections
import re
import sys


def body(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op in ("s_nop",):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "valu_cmp_sel"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    ls = [l.strip() for l in body(path, pat)]
    ins = [l.split()[0] for l in ls if l and not l.startswith((";", ".", "_Z")) and not l.endswith(":")]
    if "--after-last-mfma" in sys.argv:
        idx = max(i for i, o in enumerate(ins) if o.startswith("v_mfma"))
        ins = ins[idx + 1:]
    c = collections.Counter(classify(o) for o in ins)
    print(f"{pat}: {len(ins)} instructions", dict(c))
    top = collections.Counter(ins).most_common(28)
    print("  ", ", ".join(f"{k} {v}" for k, v in top))
    for l in ls:
        if "vgpr_count" in l or "sgpr_spill" in l or "vgpr_spill" in l or "ScratchSize" in l or "Occupancy" in l:
            print("  ", l)


if __name__ == "__main__":
    main()
