"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/tlhip.h declares, and the ctypes signature table covers exactly those symbols. No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tlhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 40
    for must in ("tl_qweight_upload", "tl_dequantize_rows", "tl_matvec_q", "tl_rmsnorm_vector", "tl_rope",
                 "tl_attention", "tl_swiglu", "tl_model_decode", "tl_model_generate_greedy", "tl_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(pkg):
    from tinyllama_cpp_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build libtlhip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(capi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} is declared in include/tlhip.h but not exported"


def test_ctypes_table_matches_header(pkg):
    from tinyllama_cpp_amd import capi
    assert sorted(list(capi.SIGNATURES) + ["tl_last_error"]) == declared_symbols()


def test_missing_library_fails_loudly(pkg, monkeypatch):
    from tinyllama_cpp_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtlhip.so")
    with pytest.raises(capi.TlError):
        capi.lib()
