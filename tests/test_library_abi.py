"""CPU: the C-ABI shared library loads and exports every symbol include/semz_gpu.h
declares; entry points that need a GPU fail loudly without one (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from decompress_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "semz_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(semz_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/semz_gpu.h but not exported"
    assert set(_lib.EXPORTS) == set(names), set(_lib.EXPORTS) ^ set(names)


def test_struct_layouts_match_header():
    # sizes the kernels and the ctypes mirror must agree on (semz_gpu.h)
    assert C.sizeof(_lib.Unit) == 96
    assert C.sizeof(_lib.Result) == 72
    assert C.sizeof(_lib.StreamInfo) == 56


def test_strerror_texts():
    # message texts of the reference's `die` sites (common.lisp:53-57 and callers)
    assert _lib.strerror(0) == "ok"
    assert "Premature end" in _lib.strerror(1)
    assert _lib.strerror(10) == "Underfull Huffman tree."
    assert _lib.strerror(11) == "Overfull Huffman tree."
    assert _lib.strerror(20) == "Block uses reserved BTYPE."
    assert _lib.strerror(170) == "Trailing data detected."
    for code in list(range(0, 2)) + list(range(10, 12)) + list(range(20, 29)) + list(range(40, 46)) + \
            list(range(50, 57)) + list(range(60, 76)) + list(range(80, 94)) + list(range(100, 108)) + \
            list(range(120, 157)) + [170, 200, 201, 202, 203, 204]:
        assert _lib.strerror(code) and "unknown status" not in _lib.strerror(code).lower(), code


def test_no_gpu_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EngineError):
        _lib.Engine(0)
