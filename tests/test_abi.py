"""CPU suite: the C-ABI library loads and exports what include/m4g.h declares;
without a GPU every compute entry point fails loudly (there is no fallback)."""
import ctypes
import os
import re

import pytest

from brotli_amd import m4

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "m4g.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m4g_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 15
    L = ctypes.CDLL(m4.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/m4g.h but not exported"
    assert set(syms) == set(m4.EXPORTS)


def test_error_strings():
    L = m4.lib()
    assert L.m4g_strerror(0) == b"ok"
    assert b"gfx950" in L.m4g_strerror(-1000001)


def test_cpu_device_is_rejected():
    """device < 0 (the survey's 'CPU oracle backend') does not exist on purpose."""
    L = m4.lib()
    h = ctypes.c_void_p()
    p = m4.Params(0, 0, 0, 0, 0, 0)
    assert L.m4g_create(ctypes.byref(p), -1, ctypes.byref(h)) == -1000001
    assert L.m4g_batch_create(ctypes.byref(p), -1, 4, 1 << 16, ctypes.byref(h)) == -1000001


def test_bad_params_rejected():
    L = m4.lib()
    h = ctypes.c_void_p()
    for bad in (m4.Params(0, 0, 9, 0, 0, 0), m4.Params(0, 0, 0, 25, 0, 0), m4.Params(-1, 0, 0, 0, 0, 0)):
        assert L.m4g_create(ctypes.byref(bad), 0, ctypes.byref(h)) == -1000002


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(m4.M4GError):
        m4.M4().find_matches(b"hello hello hello hello")
    with pytest.raises(m4.M4GError):
        m4.BatchPlan(4, 1 << 16)


def test_product_does_not_reference_oracle():
    """Nothing under brotli_amd/ or bench-independent product code may import,
    link or open anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "brotli_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "m4_ref" not in txt and "libm4oracle" not in txt, f
