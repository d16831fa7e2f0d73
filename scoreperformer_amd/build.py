"""Build the C-ABI library ``libspn.so`` (HIP kernels for gfx950) in-tree with hipcc.

``python -m scoreperformer_amd.build`` or ``scoreperformer_amd.build.build()``.
hipcc cross-compiles for gfx950 without a GPU; the ``.so`` travels to the GPU box with the tree.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libspn.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-source additions.  attention*: no SLP vectorisation -- it packs neighbouring fp32 multiplies / adds into v_pk_*_f32, which cost
# ~1.9 VALU issue slots for two results (tools/issue_probe.hip) and surround them with v_mov shuffles, in kernels bound by VALU issue
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"], "attention_dkv.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(src):
    """Content hash of everything an object depends on: compiler flags, the source and every header of csrc/ (a stale object --
    older flags, a reverted header with an older mtime -- is rebuilt, not reused)."""
    h = hashlib.sha256(" ".join([HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])).encode())
    inc = os.path.join(HERE, "..", "include")   # include/spn.h: csrc/decode_layer.hip takes its argument struct from the C-ABI header
    for path in [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")):
        with open(path, "rb") as fh:
            h.update(path.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _stale(src, obj):
    stamp = obj + ".sha256"
    if not (os.path.exists(obj) and os.path.exists(stamp)):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != _digest(src)


def _compile(name):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name.rsplit(".", 1)[0] + ".o")
    if _stale(src, obj):
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(name, []) + (["-x", "hip"] if name.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stderr[-4000:]}")
        with open(obj + ".sha256", "w") as fh:
            fh.write(_digest(src))
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, _sources()))
    link_stamp = os.path.join(OBJ, "libspn.link.sha256")
    want = hashlib.sha256("".join(open(o + ".sha256").read() for o in objs).encode()).hexdigest()
    have = open(link_stamp).read().strip() if os.path.exists(link_stamp) else ""
    if force or not os.path.exists(LIB) or have != want:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(link_stamp, "w") as fh:
            fh.write(want)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
