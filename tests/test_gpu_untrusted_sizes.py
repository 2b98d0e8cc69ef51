"""GPU: sizes that come out of the data are hints, never allocation orders.

An xz index whose records lie (sums that wrap 2^64, terabyte blocks, too-small blocks), BGZF members whose
ISIZE is wrong in either direction, block headers and .lzma headers that declare absurd sizes: the call
must succeed (rc 0), report exactly what the oracle's sequential walk reports (the reference's condition:
an index mismatch, a trailer error, a size mismatch) and deliver the same bytes before the error.
"""
import lzma

import pytest

import corpus
import xzutil
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu


def compare(engine, fmt, data, cap=1 << 23, opt=0):
    got, res, info = engine.decode_stream(fmt, data, True)         # raises on rc != 0: data errors never fail a call
    er, eo = oracle.decompress(fmt, data, True, opt=opt, cap=cap)
    assert res.status == er.status, (fmt, res.status, er.status, res.detail, er.detail)
    if er.status == 0:
        assert got == eo and res.in_end_bit == er.in_end_bit
    else:
        assert res.detail == er.detail, (fmt, res.status, res.detail, er.detail)
        assert got[:len(eo)] == eo[:len(got)]
    return got, res


def test_xz_index_that_lies(engine):
    d = corpus.x86ish(600000, seed=21)
    good = xzutil.xz_cli(d, "--block-size=131072", "--lzma2=preset=6", "--check=crc64")
    got, res = compare(engine, 8, good)
    assert res.status == 0 and got == d
    forged = [
        lambda old: [(u, 1 << 62) for u, _ in old],                          # the sum wraps to a small number
        lambda old: [(u, (1 << 63) - 1) for u, _ in old[:2]] + old[2:],      # two records that add up to 2^64 - 2
        lambda old: [old[0], (old[1][0], 1 << 40)] + old[2:],                # one terabyte block
        lambda old: [old[0], (old[1][0], old[1][1] - 1)] + old[2:],          # span one byte too small
        lambda old: [old[0], (old[1][0], 5)] + old[2:],                      # span far too small
        lambda old: [(u, s + 1) for u, s in old],                            # every span one byte too large
        lambda old: old + [(12, 0)],                                         # a record more than there are blocks
        lambda old: old[:-1],                                                # a record less
        lambda old: [(u + 4, s) for u, s in old],                            # unpadded sizes off: block positions wrong
    ]
    for k, edit in enumerate(forged):
        bad = xzutil.xz_rewrite_index(good, edit=edit)
        got, res = compare(engine, 8, bad)
        assert res.status in (130, 131), (k, res.status)                     # index disagrees with the observed blocks
        assert got == d                                                      # every block was delivered before the index


def test_bgzf_isize_that_lies(engine):
    d = corpus.text(400_000, seed=22)
    good = xzutil.bgzf(d, 50000)
    members = []
    pos = 0
    while pos < len(good):
        size = int.from_bytes(good[pos + 16:pos + 18], "little") + 1
        members.append(good[pos:pos + size])
        pos += size
    assert len(members) == 8
    for k, isize in ((3, 50000 - 1), (3, 17), (5, 0), (2, 50001), (4, 0xFFFFFFFF), (0, 1 << 31)):
        m = bytearray(members[k])
        m[-4:] = isize.to_bytes(4, "little")
        bad = b"".join(members[:k]) + bytes(m) + b"".join(members[k + 1:])
        got, res = compare(engine, 2, bad)
        assert res.status == 56 and got == d[:50000 * k]                     # "Invalid data length check", gzip.lisp:111-113
    # sixteen members that each claim 4 GiB - 1: no 64 GiB arena, the same error at the first of them
    bad = b"".join(bytes(m[:-4]) + b"\xff\xff\xff\xff" for m in members * 2)
    got, res = compare(engine, 2, bad)
    assert res.status == 56 and got == b""


def test_block_header_sizes_that_lie(engine):
    d = corpus.mixed(70000, seed=23)
    for usize in (len(d), len(d) - 1, len(d) + 1, 1, 1 << 40, (1 << 62) + 5):
        x = xzutil.xz_block_with_declared_sizes(d, usize=usize)
        got, res = compare(engine, 8, x)
        assert (res.status == 0) == (usize == len(d))
    raw_len = len(lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 6, "dict_size": 1 << 20}]))
    for csize in (raw_len, raw_len - 1, raw_len + 1, 1, 1 << 50):
        x = xzutil.xz_block_with_declared_sizes(d, csize=csize)
        got, res = compare(engine, 8, x)
        assert (res.status == 0) == (csize == raw_len)


def test_lzma_header_size_that_lies(engine):
    d = corpus.text(30000, seed=24)
    alone = bytearray(lzma.compress(d, lzma.FORMAT_ALONE))                   # size field = -1, data ends with a marker
    engine.set_option("lzma-honour-size", 1)
    try:
        for size in (len(d), len(d) - 1, len(d) + 1, 1 << 35, (1 << 63) + 9):
            alone[5:13] = size.to_bytes(8, "little")
            got, res, _ = engine.decode_stream(5, bytes(alone), True)
            er, eo, _, _ = oracle.lzma_alone(bytes(alone), honour_size=True, cap=len(d) + 64)
            assert (res.status, res.detail) == (er.status, er.detail), (size, res.status, er.status)
            assert got[:len(eo)] == eo[:len(got)]
    finally:
        engine.set_option("lzma-honour-size", 0)
