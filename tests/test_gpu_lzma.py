"""GPU parity for :raw-lzma / :lzma (lzma.lisp:809-895): SEMZ_UNIT_LZMA / SEMZ_UNIT_LZMA_WIDE units through
semz_decode_batch and the two stream formats through semz_decode_stream, against the oracle."""
import lzma
import random

import pytest

import corpus
import lzmaenc
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu


def lzma_unit(kind, off, blob, out_off, cap, lc, lp, pb, dict_size, size=None, mode=0):
    return _lib.Unit(in_off=off, in_limit=off + len(blob), out_off=out_off, out_cap=cap, kind=kind, param0=dict_size,
                     param1=lc | lp << 8 | pb << 16, param2=mode, flags=0 if size is None else _lib.UF_HAVE_OUT_SIZE,
                     size_out=size or 0)


def run_cases(engine, cases, kind=_lib.UNIT_LZMA, slot=1 << 17):
    """cases: (blob, lc, lp, pb, dict, size, mode, cap). One batch; every unit checked against the oracle."""
    data = b"".join(c[0] for c in cases)
    units, off = [], 0
    for i, (blob, lc, lp, pb, ds, size, mode, cap) in enumerate(cases):
        units.append(lzma_unit(kind, off, blob, i * slot, cap, lc, lp, pb, ds, size, mode))
        off += len(blob)
    out, res = engine.decode_batch(data, units, len(cases) * slot)
    for u, r, (blob, lc, lp, pb, ds, size, mode, cap) in zip(units, res, cases):
        er, eo = oracle.lzma1(blob, lc, lp, pb, ds, size, mode, cap=cap)
        assert r.status == er.status, (len(blob), lc, lp, pb, size, mode, cap, r.status, er.status, r.detail)
        if er.status == 0:
            assert out.raw[u.out_off:u.out_off + r.out_len] == eo
            assert r.in_end_bit - 8 * u.in_off == er.in_end_bit
        else:
            assert r.detail == er.detail
    return res


def test_liblzma_streams(engine):
    cases = []
    for i, (lc, lp, pb, n) in enumerate([(3, 0, 2, 100000), (0, 0, 0, 1), (4, 0, 4, 70000), (0, 4, 0, 30000), (2, 2, 1, 120000),
                                         (1, 3, 3, 500), (3, 0, 2, 0)]):
        d = corpus.mixed(n, seed=20 + i)
        a = lzma.compress(d, lzma.FORMAT_ALONE, filters=[{"id": lzma.FILTER_LZMA1, "preset": 5, "lc": lc, "lp": lp, "pb": pb,
                                                            "dict_size": 1 << 16}])
        cases.append((a[13:], lc, lp, pb, 1 << 16, None, 0, n + 8))
    res = run_cases(engine, cases)
    assert all(r.status == 0 for r in res)


def test_eof_modes_and_sizes(engine):
    d = corpus.mixed(5000, seed=6)
    with_marker, without = lzmaenc.encode(d, marker=True), lzmaenc.encode(d, marker=False)
    cut = next(n for n in range(4000, 4300) if oracle.lzma1(without, 3, 0, 2, 1 << 16, n, 2, cap=8000)[0].status == 89)
    cases = []
    for blob in (with_marker, without):
        for size in (None, 5000, cut, cut + 1, cut + 2, 6000, 0):
            for mode in (0, 1, 2):
                cases.append((blob, 3, 0, 2, 1 << 16, size, mode, 8000))
    cases.append((with_marker, 3, 0, 2, 1 << 16, None, 0, 4999))                   # output span too small
    cases.append((with_marker, 3, 0, 2, 1 << 16, None, 0, 5000))                   # exactly full, then the marker
    cases.append((without, 3, 0, 2, 1 << 16, 5000, 0, 4000))                       # declared size beyond the span
    cases.append((with_marker, 3, 0, 2, 64, None, 0, 8000))                        # window smaller than the matches
    cases.append((b"\1" + with_marker[1:], 3, 0, 2, 1 << 16, None, 0, 8000))
    cases.append((b"\0\xff\xff\xff\xff" + with_marker[5:], 3, 0, 2, 1 << 16, None, 0, 8000))
    cases.append((with_marker, 4, 0, 2, 1 << 16, None, 0, 8000))                   # wrong properties: garbage or an error
    cases.append((with_marker, 3, 0, 0, 1 << 16, None, 0, 8000))
    res = run_cases(engine, cases, slot=8192)
    assert {r.status for r in res} >= {0, 1, 87, 88, 89, 90, 92, 200}


def test_corrupt_and_truncated(engine):
    rng = random.Random(31)
    d = corpus.mixed(40000, seed=32)
    base = lzma.compress(d, lzma.FORMAT_ALONE, filters=[{"id": lzma.FILTER_LZMA1, "preset": 3, "dict_size": 1 << 15}])[13:]
    cases = [(base[:cut], 3, 0, 2, 1 << 15, None, 0, 48000) for cut in (0, 1, 4, 5, 6, len(base) // 2, len(base) - 1)]
    for k in range(80):
        b = bytearray(base)
        for _ in range(rng.choice([1, 1, 2, 4])):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        cases.append((bytes(b), 3, 0, 2, (1 << 15) if k % 3 else 2048, None if k % 2 else 40000, k % 3, 48000))
    run_cases(engine, cases, slot=49152)


def test_wide_literal_contexts(engine):
    cases = []
    for i, (lc, lp, pb) in enumerate([(8, 4, 4), (5, 0, 2), (8, 0, 0), (3, 4, 1), (3, 0, 2)]):
        d = corpus.mixed(12000, seed=lc * 5 + lp)
        cases.append((lzmaenc.encode(d, lc, lp, pb, 1 << 16, marker=bool(i & 1)), lc, lp, pb, 1 << 16,
                      None if i & 1 else 12000, 0, 12008))
    res = run_cases(engine, cases, kind=_lib.UNIT_LZMA_WIDE, slot=16384)
    assert all(r.status == 0 for r in res)
    # the LDS kernel refuses wide contexts instead of decoding them wrongly
    blob, lc, lp, pb = cases[0][:4]
    out, res = engine.decode_batch(blob, [lzma_unit(_lib.UNIT_LZMA, 0, blob, 0, 12008, lc, lp, pb, 1 << 16, 12000)], 12008)
    assert res[0].status == 201


def compare_stream(engine, fmt, data, all_members=True, cap=1 << 22, opt=0):
    got, res, info = engine.decode_stream(fmt, data, all_members)
    er, eo = oracle.decompress(fmt, data, all_members, opt=opt, cap=cap)
    assert res.status == er.status, (fmt, len(data), res.status, er.status, res.detail, er.detail)
    if er.status == 0:
        assert got == eo and res.in_end_bit == er.in_end_bit
    return got, res, info


def test_lzma_alone_streams(engine):
    d = corpus.text(400000, seed=33)
    a = lzma.compress(d, lzma.FORMAT_ALONE, preset=6)
    got, res, info = compare_stream(engine, 5, a)
    assert got == d and list(info.first_header[:5]) == [3, 0, 2, 1 << 23, 0]
    compare_stream(engine, 5, a + b"\0")
    got, res, _ = compare_stream(engine, 5, a + b"tail", all_members=False)
    assert got == d and res.in_end_bit == 8 * len(a)
    for cut in (0, 1, 12, 13, 18, len(a) // 2, len(a) - 1):
        compare_stream(engine, 5, a[:cut])
    compare_stream(engine, 5, bytes([225]) + a[1:])
    # output much larger than the first guess: the span is retried
    z = lzma.compress(bytes(8_000_000), lzma.FORMAT_ALONE, preset=1)
    got, res, _ = engine.decode_stream(5, z, True)
    assert res.status == 0 and len(got) == 8_000_000 and not any(got[::4099])
    # declared size without a marker: as written the reference decodes on past the data; the option honours the size
    small = corpus.mixed(20000, seed=34)
    nm = lzmaenc.alone(small, marker=False, declare_size=True)
    _, res, info = compare_stream(engine, 5, nm)
    assert res.status != 0 and list(info.first_header[4:7]) == [1, 20000, 0]
    engine.set_option("lzma-honour-size", 1)
    try:
        got, res, _ = compare_stream(engine, 5, nm, opt=1)
        assert res.status == 0 and got == small
        compare_stream(engine, 5, lzmaenc.alone(small, marker=True, declare_size=True), opt=1)
        wide = lzmaenc.alone(small[:6000], lc=7, lp=2, pb=0, marker=False, declare_size=True)
        got, res, _ = compare_stream(engine, 5, wide, opt=1)
        assert res.status == 0 and got == small[:6000]
    finally:
        engine.set_option("lzma-honour-size", 0)


def test_raw_lzma_stream_options(engine):
    d = corpus.mixed(50000, seed=35)
    raw = lzmaenc.encode(d[:9000], 2, 1, 3, 1 << 14, marker=False)
    for k, v in (("lc", 2), ("lp", 1), ("pb", 3), ("window-size", 1 << 14), ("decompressed-size", 9000), ("eof-mode", 2)):
        engine.set_option(k, v)
    got, res, _ = engine.decode_stream(4, raw, True)
    er, eo = oracle.lzma1(raw, 2, 1, 3, 1 << 14, 9000, 2, cap=9008)
    assert res.status == er.status == 0 and got == eo == d[:9000] and res.in_end_bit == er.in_end_bit
    got, res, _ = engine.decode_stream(4, raw + b"x", True)
    assert res.status == 170
    engine.set_option("eof-mode", 1)
    _, res, _ = engine.decode_stream(4, raw, True)
    assert res.status == oracle.lzma1(raw, 2, 1, 3, 1 << 14, 9000, 1, cap=9008)[0].status == 87
    engine.set_option("decompressed-size", -1)
    engine.set_option("eof-mode", 0)
    a = lzma.compress(d, lzma.FORMAT_ALONE, filters=[{"id": lzma.FILTER_LZMA1, "preset": 2, "lc": 2, "lp": 1, "pb": 3,
                                                        "dict_size": 1 << 14}])
    got, res, _ = engine.decode_stream(4, a[13:], True)
    assert res.status == 0 and got == d
    with pytest.raises(_lib.EngineError):
        engine.set_option("lc", 9)
