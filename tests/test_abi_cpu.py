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
    assert handle.spn_abi_version() == 10


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


def test_library_owns_no_memory_and_reads_no_environment():
    """include/spn.h's contract: no allocation, no device/stream synchronisation, no getenv anywhere in csrc/."""
    banned = ("hipMalloc", "hipFree", "hipDeviceSynchronize", "hipStreamSynchronize", "hipEventSynchronize", "getenv")
    csrc = os.path.join(ROOT, "scoreperformer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".h")):
            text = open(os.path.join(csrc, f)).read()
            if f == "comm.cpp":   # the one sanctioned wait: spn_comm_destroy drains ITS OWN communication stream before teardown
                head, sep, tail = text.partition('extern "C" int spn_comm_destroy')
                assert sep and tail.count("hipStreamSynchronize") == 1 and "hipStreamSynchronize" not in head
                text = head + tail.replace("hipStreamSynchronize", "", 1)
            for word in banned:
                assert word not in text, f"{f} uses {word}"


def test_tuning_table_round_trip():
    """spn_set_tuning / spn_get_tuning are host-only: every knob of csrc/tuning.h is enumerable, settable and readable."""
    from scoreperformer_amd import build, lib
    build.build()
    handle = ctypes.CDLL(lib.LIB_PATH)
    handle.spn_tuning_name.restype = ctypes.c_char_p
    names = [handle.spn_tuning_name(ctypes.c_int(i)).decode() for i in range(handle.spn_tuning_count())]
    assert "attn_band" in names and "gemm_variant" in names and len(set(names)) == len(names)
    out = ctypes.c_double(0)
    assert handle.spn_get_tuning(b"attn_band", ctypes.byref(out)) == 0 and out.value == 30.0
    assert handle.spn_set_tuning(b"attn_band", ctypes.c_double(0.0)) == 0
    assert handle.spn_get_tuning(b"attn_band", ctypes.byref(out)) == 0 and out.value == 0.0
    assert handle.spn_set_tuning(b"attn_band", ctypes.c_double(30.0)) == 0
    assert handle.spn_set_tuning(b"no_such_knob", ctypes.c_double(1.0)) != 0


def test_gemm_workspace_query_is_pure():
    """spn_gemm_workspace_bytes: 0 for bf16 outputs and for big tile grids, > 0 for the weight-gradient shapes (no GPU needed)."""
    from scoreperformer_amd import build, lib
    build.build()
    handle = ctypes.CDLL(lib.LIB_PATH)
    handle.spn_gemm_workspace_bytes.restype = ctypes.c_size_t
    q = lambda M, N, K, flags: handle.spn_gemm_workspace_bytes(ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(K), ctypes.c_int(flags), ctypes.c_int(1))
    assert q(131072, 4096, 512, 0) == 0 and q(131072, 512, 2048, 4) == 0
    need = q(4096, 512, 131072, 1 | 2 | 4 | 8)
    assert need > 0 and need % (4096 * 512 * 4) == 0
    assert q(512, 2048, 131072, 1 | 2 | 4) > 0


def test_dec_pair_groups_is_a_host_function():
    """spn_dec_pair_groups (include/spn.h): workgroups of the persistent decoder layer-pair launch -- h * S attention + ceil(d / 16)
    projection + ceil(inner / 32) feed-forward workgroups, all resident at once (<= 256 CUs); 0 for shapes the launch does not take."""
    from scoreperformer_amd import build, lib
    build.build()
    handle = ctypes.CDLL(lib.LIB_PATH)
    assert handle.spn_dec_pair_groups(512, 8, 1, 2048, 16) == 128 + 32 + 64     # C5 decoder
    assert handle.spn_dec_pair_groups(512, 8, 8, 2048, 16) == 224               # multi-head keys / values
    assert handle.spn_dec_pair_groups(128, 2, 1, 512, 16) == 32 + 8 + 16
    assert handle.spn_dec_pair_groups(1024, 16, 1, 4096, 16) == 0               # wider than the kernel's register plan
    assert handle.spn_dec_pair_groups(512, 8, 1, 2048, 32) == 0                 # more splits than the merge holds


def test_decode_record_layouts_match_the_header():
    """The ctypes mirrors of spn_dec_pair_args / spn_dec_chain_ext have the size the compiled header gives them (no GPU needed)."""
    import ctypes
    from scoreperformer_amd import ops
    from scoreperformer_amd.lib import load
    handle = load()
    assert handle.spn_dec_struct_size(0) == ctypes.sizeof(ops.DecPairArgs)
    assert handle.spn_dec_struct_size(1) == ctypes.sizeof(ops.DecChainExt)
