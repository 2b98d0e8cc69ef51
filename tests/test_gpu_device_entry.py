"""GPU: semz_decode_batch_device checks every unit's spans on the device (SURVEY.md section 5:
"every device-side index must be bounds-checked against unit in_len / out_cap").

A unit whose spans leave the buffers of the call gets SEMZ_E_BAD_UNIT and nothing is read or
written through it; the good units of the same batch decode as usual (oracle-checked).
"""
import bz2
import ctypes as C
import lzma
import zlib

import pytest

import corpus
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu

BAD_UNIT = 203


def run_device(engine, kind, data, units, out_cap, in_len=None):
    n = len(units)
    uarr = (_lib.Unit * n)(*units)
    d_in = engine.dev_alloc(len(data) + 16)
    d_units = engine.dev_alloc(C.sizeof(uarr))
    d_out = engine.dev_alloc(out_cap + 16)
    d_res = engine.dev_alloc(n * C.sizeof(_lib.Result))
    try:
        engine.upload(d_in, data)
        engine.upload(d_units, bytes(uarr))
        engine.upload(d_out, b"\xEE" * (out_cap + 16))
        engine.upload(d_res, b"\xFF" * (n * C.sizeof(_lib.Result)))
        engine.decode_batch_device(kind, d_in, len(data) if in_len is None else in_len, d_units, n, d_out, out_cap, d_res)
        engine.sync()
        out = engine.download(d_out, out_cap + 16)
        res = (_lib.Result * n).from_buffer_copy(engine.download(d_res, n * C.sizeof(_lib.Result)))
        return out, list(res)
    finally:
        for p in (d_in, d_units, d_out, d_res):
            engine.dev_free(p)


def test_deflate_family_bad_spans(engine):
    d = corpus.text(20000, seed=1)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = co.compress(d) + co.flush()
    n, cap = len(blob), len(d)
    good = dict(in_off=0, in_limit=n, out_off=0, out_cap=cap, kind=_lib.UNIT_GZIP)
    units = [
        _lib.Unit(**good),
        _lib.Unit(**{**good, "in_limit": n + 1, "out_off": cap}),                 # reads past the input
        _lib.Unit(**{**good, "in_off": n, "in_limit": n - 1, "out_off": cap}),      # in_off > in_limit
        _lib.Unit(**{**good, "out_off": cap + 1}),                                   # span leaves the output by one byte
        _lib.Unit(**{**good, "out_off": 2 ** 63, "out_cap": 2 ** 63}),              # out_off + out_cap wraps
        _lib.Unit(**{**good, "out_off": _lib.OUT_AUTO}),                            # AUTO is for bzip2 blocks
        _lib.Unit(**{**good, "out_off": cap, "prefix_off": n - 2, "prefix_len": 3}),  # prefix leaves the input
        _lib.Unit(**{**good, "out_off": cap, "kind": _lib.UNIT_XZ_BLOCK}),          # other family
        _lib.Unit(**{**good, "out_off": cap}),
    ]
    out, res = run_device(engine, _lib.UNIT_GZIP, blob, units, 2 * cap)
    assert [r.status for r in res] == [0] + [BAD_UNIT] * 7 + [0]
    er, eo, _ = oracle.gzip_member(blob, cap)
    assert er.status == 0 and out[:cap] == eo == d and out[cap:2 * cap] == d
    assert out[2 * cap:] == b"\xEE" * 16                                             # nothing beyond the call's out_cap
    assert bytes(res[0].check[:4]) == bytes(er.check[:4])


def test_bzip2_bad_spans(engine):
    d = corpus.text(50000, seed=2)
    blob = bz2.compress(d, 1)
    n = len(blob)
    good = dict(in_off=4, in_limit=n, out_off=_lib.OUT_AUTO, kind=_lib.UNIT_BZIP2_BLOCK, param0=1)
    units = [
        _lib.Unit(**good),
        _lib.Unit(**{**good, "in_limit": n + 100}),
        _lib.Unit(**{**good, "in_bit_off": 8}),
        _lib.Unit(**{**good, "out_off": 60000, "out_cap": 60000}),                  # fixed span outside the output
        _lib.Unit(**good),
    ]
    out, res = run_device(engine, _lib.UNIT_BZIP2_BLOCK, blob, units, 2 * len(d))
    assert [r.status for r in res] == [0, BAD_UNIT, BAD_UNIT, BAD_UNIT, 0]
    assert out[:len(d)] == d and out[len(d):2 * len(d)] == d
    assert res[4].out_off == len(d)                                                  # bad units take no room in the packing


def test_xz_bad_spans(engine):
    d = corpus.mixed(40000, seed=3)
    blob = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 6}])
    n, cap = len(blob), len(d)
    good = dict(in_off=0, in_limit=n, out_off=0, out_cap=cap, kind=_lib.UNIT_LZMA2, param0=1 << 23)
    units = [
        _lib.Unit(**good),
        _lib.Unit(**{**good, "in_limit": n + 1}),
        _lib.Unit(**{**good, "out_cap": cap + 1}),
        _lib.Unit(**{**good, "kind": _lib.UNIT_GZIP}),
    ]
    out, res = run_device(engine, _lib.UNIT_LZMA2, blob, units, cap)
    assert [r.status for r in res] == [0, BAD_UNIT, BAD_UNIT, BAD_UNIT]
    assert out[:cap] == d and out[cap:] == b"\xEE" * 16


def test_call_level_in_len_bounds_units(engine):
    # the same unit table is fine with the real in_len and rejected when the call says the input is shorter
    d = corpus.text(3000, seed=4)
    blob = zlib.compress(d, 6)
    u = _lib.Unit(in_off=0, in_limit=len(blob), out_off=0, out_cap=len(d), kind=_lib.UNIT_ZLIB)
    _, res = run_device(engine, _lib.UNIT_ZLIB, blob, [u], len(d))
    assert res[0].status == 0
    _, res = run_device(engine, _lib.UNIT_ZLIB, blob, [u], len(d), in_len=len(blob) - 1)
    assert res[0].status == BAD_UNIT
