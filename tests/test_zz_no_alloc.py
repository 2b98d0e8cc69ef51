"""GPU: semz_decode_batch_device_noalloc (the entry point that allocates nothing) against
semz_decode_batch_device, through the C ABI.

Bar (byte work): identical output bytes and identical semz_result records (status, detail,
consumed bits, length, check value) on two calls in a row, the second with a smaller input,
so that whatever the first call left in the library's static scratch array cannot leak into
the second; and every unit's result equals the CPU oracle's (status, detail of a failure;
bytes, consumed bits and check value of a success), so the pair cannot be wrong together.
The limits of the entry point are checked as API errors.
"""
import ctypes as C
import random
import zlib

import pytest

import corpus
from decompress_amd import _lib
from test_gpu_deflate import oracle_unit

pytestmark = pytest.mark.gpu


def build_batch(n_units, seed, corrupt_every=0):
    """n_units gzip/zlib/deflate units back to back; returns (data, [Unit], out_cap)."""
    rng = random.Random(seed)
    data = bytearray()
    units = []
    out_off = 0
    for i in range(n_units):
        fmt = i % 3
        n = rng.choice([0, 1, 100, 4096, 30000, 70000])
        d = [corpus.text, corpus.mixed, corpus.repetitive][i % 3](n, seed=seed * 1000 + i) if n else b""
        wbits = {0: -15, 1: 15, 2: 31}[fmt]
        co = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, wbits)
        blob = bytearray(co.compress(d) + co.flush())
        if corrupt_every and i % corrupt_every == corrupt_every - 1 and len(blob) > 12:
            blob[rng.randrange(len(blob))] ^= 1 << rng.randrange(8)          # reject paths must agree as well
        data += b"\x55" * rng.randrange(0, 4)
        units.append(_lib.Unit(in_off=len(data), in_limit=len(data) + len(blob), out_off=out_off, out_cap=n + 300,
                               kind=(_lib.UNIT_DEFLATE, _lib.UNIT_ZLIB, _lib.UNIT_GZIP)[fmt]))
        data += blob
        out_off += n + 300 + rng.randrange(0, 5)
    return bytes(data), units, out_off + 16


def run_device(engine, call, data, units, out_cap):
    n = len(units)
    uarr = (_lib.Unit * n)(*units)
    d_in = engine.dev_alloc(len(data) + 16)
    d_units = engine.dev_alloc(C.sizeof(uarr))
    d_out = engine.dev_alloc(out_cap)
    d_res = engine.dev_alloc(n * C.sizeof(_lib.Result))
    try:
        engine.upload(d_in, data)
        engine.upload(d_units, bytes(uarr))
        engine.upload(d_out, b"\xEE" * out_cap)
        engine.upload(d_res, b"\xFF" * (n * C.sizeof(_lib.Result)))
        call(_lib.UNIT_GZIP, d_in, len(data), d_units, n, d_out, out_cap, d_res)
        engine.sync()
        out = engine.download(d_out, out_cap)
        res = (_lib.Result * n).from_buffer_copy(engine.download(d_res, n * C.sizeof(_lib.Result)))
        return out, list(res)
    finally:
        for p in (d_in, d_units, d_out, d_res):
            engine.dev_free(p)


def same_results(a, b):
    for x, y in zip(a, b):
        assert (x.status, x.detail, x.in_end_bit, x.out_len, x.out_off, x.check_len, x.aux) == \
               (y.status, y.detail, y.in_end_bit, y.out_len, y.out_off, y.check_len, y.aux)
        assert bytes(x.check) == bytes(y.check)


def test_noalloc_matches_device_entry_twice(engine):
    # first call: 300 units; second call: 40 units (smaller input, smaller table)
    for n_units, seed in ((300, 1), (40, 2)):
        data, units, out_cap = build_batch(n_units, seed, corrupt_every=7)
        want_out, want = run_device(engine, engine.decode_batch_device, data, units, out_cap)
        got_out, got = run_device(engine, engine.decode_batch_device_noalloc, data, units, out_cap)
        assert any(r.status == 0 for r in want) and any(r.status != 0 for r in want)
        same_results(got, want)
        for u, r in zip(units, want):
            assert got_out[u.out_off:u.out_off + r.out_len] == want_out[u.out_off:u.out_off + r.out_len]
        # bytes outside the units' spans are untouched by both
        assert got_out == want_out
        # ... and the oracle says the same, unit by unit
        fmt_of = {_lib.UNIT_DEFLATE: 0, _lib.UNIT_ZLIB: 1, _lib.UNIT_GZIP: 2}
        for u, r in zip(units, got):
            er, eo = oracle_unit(fmt_of[u.kind], data[u.in_off:u.in_limit], u.out_cap)
            assert r.status == er.status, (r.status, r.detail, er.status, er.detail)
            if er.status == 0:
                assert got_out[u.out_off:u.out_off + r.out_len] == eo
                assert r.in_end_bit - 8 * u.in_off == er.in_end_bit
                assert bytes(r.check[:r.check_len]) == bytes(er.check[:er.check_len])
            else:
                assert r.detail == er.detail, (r.status, r.detail, er.detail)


def test_noalloc_limits(engine):
    data, units, out_cap = build_batch(4, 3)
    d = engine.dev_alloc(4096)
    try:
        # more units than the static table holds: refused with a message, nothing launched
        with pytest.raises(_lib.EngineError, match="at most 65536 units"):
            engine.decode_batch_device_noalloc(_lib.UNIT_GZIP, d, 4096, d, 65537, d, 4096, d)
        # families that need per-batch scratch are not served by this entry point
        for kind in (_lib.UNIT_BZIP2_BLOCK, _lib.UNIT_XZ_BLOCK, _lib.UNIT_LZMA2):
            with pytest.raises(_lib.EngineError, match="deflate / zlib / gzip units only"):
                engine.decode_batch_device_noalloc(kind, d, 4096, d, 1, d, 4096, d)
    finally:
        engine.dev_free(d)
