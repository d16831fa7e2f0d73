"""CPU: the C-ABI library builds, loads, and exports every symbol that include/spn.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "spn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from scoreperformer_amd import build, lib
    build.build()
    handle = ctypes.CDLL(lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(handle, s)]
    assert not missing, missing
    assert handle.spn_abi_version() == 1


def test_python_bindings_call_only_declared_symbols():
    src = open(os.path.join(ROOT, "scoreperformer_amd", "ops.py")).read()
    used = set(re.findall(r'call\("(spn_[a-z0-9_]+)"', src))
    assert used <= set(declared_symbols()), used - set(declared_symbols())


def test_product_has_no_cpu_fallback():
    """Ops refuse CPU tensors loudly instead of silently computing on the host."""
    import torch
    from scoreperformer_amd import ops
    from scoreperformer_amd.lib import SpnError
    with pytest.raises(SpnError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
    # and the product never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scoreperformer_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
