"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/tlhip.h declares, and the ctypes signature table covers exactly those symbols; the same for the measurement /
diagnostic hooks of include/tlhip_debug.h (kept out of the boundary header) and the tools library. No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="tlhip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 40
    for must in ("tl_qweight_upload", "tl_dequantize_rows", "tl_matvec_q", "tl_rmsnorm_vector", "tl_rope",
                 "tl_attention", "tl_swiglu", "tl_model_decode", "tl_model_generate_greedy", "tl_last_error"):
        assert must in syms
    # the boundary header holds no bench / debug / experiment hooks (those live in tlhip_debug.h)
    assert not [s for s in syms if s.startswith(("tl_bench_", "tl_debug_", "tl_chain_"))]
    for hook in ("tl_stream_matvec", "tl_model_profile_decode", "tl_model_bench_kernel"):
        assert hook not in syms


def test_library_exports_every_declared_symbol(pkg):
    from tinyllama_cpp_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build libtlhip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(capi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} is declared in include/tlhip.h but not exported"
    for s in capi.DEBUG_SIGNATURES:
        assert hasattr(L, s), f"{s} is declared in include/tlhip_debug.h but not exported"
    for s in capi.TOOLS_SIGNATURES:
        assert not hasattr(L, s), f"{s} belongs to libtlhip_tools.so, not to the product library"
    assert os.path.exists(capi.TOOLS_LIB_PATH), "build libtlhip_tools.so first (make -C tinyllama.cpp_amd/csrc)"
    T = ctypes.CDLL(capi.TOOLS_LIB_PATH)
    for s in capi.TOOLS_SIGNATURES:
        assert hasattr(T, s), f"{s} is not exported by libtlhip_tools.so"


def test_ctypes_table_matches_header(pkg):
    from tinyllama_cpp_amd import capi
    assert sorted(list(capi.SIGNATURES) + ["tl_last_error"]) == declared_symbols()
    assert sorted(list(capi.DEBUG_SIGNATURES) + list(capi.TOOLS_SIGNATURES)) == declared_symbols("tlhip_debug.h")


def test_missing_library_fails_loudly(pkg, monkeypatch):
    from tinyllama_cpp_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtlhip.so")
    with pytest.raises(capi.TlError):
        capi.lib()
