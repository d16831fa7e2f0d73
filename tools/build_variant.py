"""Variant build of libspn.so for same-box A/B runs: ONE source recompiled with extra flags, linked with the shipped objects.
    python tools/build_variant.py gemm.hip nt_spn.so -DSPN_ST_AUX=2       ->  tools/_bin/nt_spn.so   (load it with SPN_LIB=tools/_bin/nt_spn.so)
(names matching libspn_*.so are not shipped to the GPU box: .gpurunignore)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scoreperformer_amd import build as B  # noqa: E402

src, out, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
os.makedirs(os.path.join(ROOT, "tools", "_bin"), exist_ok=True)
obj = os.path.join("/tmp", "variant_" + src.rsplit(".", 1)[0] + ".o")
cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(B.CSRC, src), "-o", obj]
subprocess.run(cmd, check=True)
objs = [obj if f == src else os.path.join(B.OBJ, f.rsplit(".", 1)[0] + ".o") for f in B._sources()]
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "tools", "_bin", out)] + objs, check=True)
print("built tools/_bin/" + out)
