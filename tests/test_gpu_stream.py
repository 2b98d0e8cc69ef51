"""GPU parity: semz_decode_stream (decompress / decompress-all, interface.lisp:149-208) vs the oracle's
mirror of the same API, for every container format, valid and broken inputs."""
import bz2
import gzip
import json
import lzma
import os
import random
import zlib

import pytest

import corpus
import xzutil
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FMT = {"deflate": 0, "zlib": 1, "gzip": 2, "bzip2": 3, "xz": 8}


def compare(engine, fmt, data, all_members=True, cap=1 << 23, opt=0):
    if fmt == 6:
        engine.set_option("window-size", opt)
    got, res, info = engine.decode_stream(fmt, data, all_members)
    er, eo = oracle.decompress(fmt, data, all_members, opt=opt, cap=cap)
    assert res.status == er.status, (fmt, len(data), res.status, er.status, res.detail, er.detail)
    if er.status == 0:
        assert got == eo
        assert res.in_end_bit == er.in_end_bit
    else:
        assert got[:len(eo)] == eo[:len(got)]            # whatever both delivered before the error agrees
    return got, res, info


def test_reference_vectors(engine):
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        gold = json.load(f)
    for vec in gold["vectors"]:
        if "dictionary" in vec:
            continue
        data = bytes(vec["input"])[vec.get("start", 0):vec.get("end")]
        got, res, info = compare(engine, FMT[vec["format"]], data, vec["all_members"])
        if "error" in vec:
            assert res.status not in (0, 1)
        else:
            assert res.status == 0 and got == bytes(vec["output"])
        if vec["format"] == "zlib" and "header" in vec:
            assert list(info.first_header[:3]) == [32768, 2, 0]
        if vec["format"] == "gzip" and "header" in vec:
            assert data[info.first_header[4]:info.first_header[4] + info.first_header[5]] == b"foo"


def test_gzip_streams(engine):
    parts = [corpus.text(30000, seed=1), b"", corpus.random_bytes(5000, seed=2), corpus.mixed(120000, seed=3)]
    data = b"".join(gzip.compress(p, 6) for p in parts)
    got, res, info = compare(engine, 2, data)
    assert got == b"".join(parts) and info.nmembers == 4
    got, res, _ = compare(engine, 2, data, all_members=False)
    assert got == parts[0]
    compare(engine, 2, data + b"junk")                           # garbage after the last member
    compare(engine, 2, data[:-3])                                # truncated trailer
    # BGZF: members carry their size, all decoded in one batch
    d = corpus.text(1_000_000, seed=4)
    b = xzutil.bgzf(d, 65280)
    got, res, info = compare(engine, 2, b)
    assert got == d and info.nmembers == 16
    bad = bytearray(b); bad[len(b) // 2] ^= 0x10
    compare(engine, 2, bytes(bad))
    compare(engine, 2, b + gzip.compress(b"unindexed tail"))


def test_zlib_and_deflate_streams(engine):
    d = corpus.mixed(70000, seed=5)
    z = zlib.compress(d, 9)
    compare(engine, 1, z)
    compare(engine, 1, z + b"\1\2\3")                            # trailing data (example.lisp:29-30)
    compare(engine, 1, z + b"\1\2\3", all_members=False)
    raw = zlib.compress(d, 6)[2:-4]
    compare(engine, 0, raw)
    compare(engine, 0, raw + raw)                                # deflate has no multi-member support
    compare(engine, 0, b"")


def test_bzip2_streams(engine):
    a, b = corpus.text(450000, seed=6), corpus.repetitive(100000, seed=7)
    data = bz2.compress(a, 2) + bz2.compress(b, 9)
    got, res, info = compare(engine, 3, data)
    assert got == a + b and info.nmembers == 2 and info.first_header[0] == 2
    got, _, _ = compare(engine, 3, data, all_members=False)
    assert got == a
    compare(engine, 3, bz2.compress(b"", 5))
    compare(engine, 3, data[:-1])
    compare(engine, 3, data[:len(data) // 3])
    compare(engine, 3, data + b"BZh")
    compare(engine, 3, data + b"xyz!")
    rng = random.Random(8)
    for _ in range(12):
        bad = bytearray(data)
        bad[rng.randrange(4, len(bad))] ^= 1 << rng.randrange(8)
        compare(engine, 3, bytes(bad))
    # a block magic inside a literal payload cannot be told from a block start by the scan
    magic = (0x314159265359).to_bytes(6, "big")
    tricky = bz2.compress(corpus.random_bytes(3000, seed=9) + magic * 3 + corpus.random_bytes(3000, seed=10), 1)
    compare(engine, 3, tricky)


def test_xz_streams(engine):
    d = corpus.x86ish(700000, seed=11)
    data = xzutil.xz_cli(d, "--block-size=131072", "--x86", "--lzma2=preset=6", "--check=sha256")
    got, res, info = compare(engine, 8, data)
    assert got == d and info.first_header[0] == 10
    two = data + b"\0" * 4 + xzutil.xz_py(b"second member", lzma.CHECK_CRC32) + b"\0" * 8
    got, _, info = compare(engine, 8, two)
    assert got == d + b"second member" and info.nmembers == 2
    got, _, _ = compare(engine, 8, two, all_members=False)
    assert got == d
    compare(engine, 8, data + b"\0" * 3)                         # padding not a multiple of four
    compare(engine, 8, data + b"garbage!")
    compare(engine, 8, data[:-5])
    compare(engine, 8, data[:len(data) // 2])
    rng = random.Random(12)
    for _ in range(16):
        bad = bytearray(data)
        where = rng.choice([rng.randrange(0, 12), rng.randrange(12, len(bad) - 60), rng.randrange(len(bad) - 60, len(bad))])
        bad[where] ^= 1 << rng.randrange(8)
        compare(engine, 8, bytes(bad))
    compare(engine, 8, xzutil.xz_py(b"", lzma.CHECK_CRC64))     # a member without blocks


def test_lzma2_streams(engine):
    """:raw-lzma2 (window size from the caller) and :lzma2 (dictionary size byte first), lzma2.lisp:205-227."""
    d = corpus.mixed(300000, seed=13)
    raw = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 5, "dict_size": 1 << 18}])
    got, res, _ = compare(engine, 6, raw, opt=1 << 18)
    assert got == d and res.status == 0
    compare(engine, 6, raw, opt=4096)                            # distances beyond the caller's window are rejected
    compare(engine, 6, raw + b"tail", opt=1 << 18)                         # no multi-member support: trailing data
    got, res, _ = compare(engine, 6, raw + b"tail", all_members=False, opt=1 << 18)
    assert got == d and res.in_end_bit == 8 * len(raw)
    compare(engine, 6, raw[:len(raw) // 2], opt=1 << 18)
    compare(engine, 6, b"", opt=1 << 18)
    for o in (0, 1, 14, 15, 18, 39, 40):                         # 2^18 needs byte 14; smaller windows fail on distance
        got, res, info = compare(engine, 7, bytes([o]) + raw)
        want = 0xFFFFFFFF if o == 40 else (2 | (o & 1)) << (11 + o // 2)
        assert info.first_header[0] == want
        assert (res.status == 0 and got == d) == (want >= 1 << 18)
    for o in (41, 200, 255):
        _, res, _ = compare(engine, 7, bytes([o]) + raw)
        assert res.status == 101 and res.detail == o
    compare(engine, 7, b"")
    compare(engine, 7, bytes([20]))
    compare(engine, 7, bytes([20]) + raw + b"x")
    rng = random.Random(14)
    for _ in range(12):
        bad = bytearray(bytes([16]) + raw)
        bad[rng.randrange(1, len(bad))] ^= 1 << rng.randrange(8)
        compare(engine, 7, bytes(bad))


def test_plain_gzip_members_with_lookalike_headers(engine):
    """Concatenated members without an index are decoded as one speculative batch (every 1F 8B 08 is a candidate):
    look-alike headers inside member data, members the hypothesis cannot size, damage and trailing data must
    all come out as the sequential walk of the reference has them (interface.lisp:91-133)."""
    rng = random.Random(40)
    fake = b"\x1f\x8b\x08\x00" + bytes(14)
    parts = []
    for i in range(12):
        body = corpus.mixed(rng.randrange(0, 40000), seed=50 + i)
        if i % 3 == 0:
            body = body[:100] + fake * 3 + body[100:] + fake          # ends up verbatim in stored blocks at level 0
        parts.append(body)
    members = [gzip.compress(p, 0 if i % 3 == 0 else 6) for i, p in enumerate(parts)]
    data = b"".join(members)
    got, res, info = compare(engine, 2, data)
    assert got == b"".join(parts) and info.nmembers == 12 and res.status == 0
    got, _, _ = compare(engine, 2, data, all_members=False)
    assert got == parts[0]
    compare(engine, 2, data + fake)                                   # a header and nothing else
    compare(engine, 2, data + b"\x1f\x8b\x08")                        # truncated header
    compare(engine, 2, data[:-9])                                     # last member without its trailer
    for k in (1, 5, 11):                                              # damage inside member k
        off = sum(len(m) for m in members[:k]) + len(members[k]) // 2
        bad = bytearray(data)
        bad[off] ^= 0x40
        compare(engine, 2, bytes(bad))
    wrong = bytearray(data)                                           # ISIZE of member 3 wrong: its span is mis-sized
    off = sum(len(m) for m in members[:4]) - 4
    wrong[off] ^= 0xFF
    compare(engine, 2, bytes(wrong))
    wrong[off + 3] ^= 0x7F                                            # and implausibly large
    compare(engine, 2, bytes(wrong))
