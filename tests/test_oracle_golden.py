"""CPU: the oracle (oracle/*.c, a C restatement of the reference) against every
known-answer vector the reference ships (tests/golden/reference_vectors.json,
typed in from example.lisp and README.org by tests/golden/make_vectors.py) and
against independent decoders for valid streams (CPython zlib / bz2 / lzma,
hashlib).  This is what pins the oracle; the GPU parity tests then compare the
HIP path with the pinned oracle.
"""
import bz2
import hashlib
import json
import lzma
import os
import random
import zlib

import pytest

import corpus
import xzutil
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    GOLD = json.load(f)

FMT = {"deflate": 0, "zlib": 1, "gzip": 2, "bzip2": 3, "raw-lzma2": 6, "xz": 8}


@pytest.mark.parametrize("vec", GOLD["vectors"], ids=[v["id"] for v in GOLD["vectors"]])
def test_reference_vector(vec):
    data = bytes(vec["input"])[vec.get("start", 0):vec.get("end")]
    if "dictionary" in vec:
        r, out, h = oracle.zlib_member(data, True, bytes(vec["dictionary"]), 64)
        assert r.status == 0 and out == bytes(vec["output"])
        assert h.dict_id == vec["header"]["dictionary"] and h.level == vec["header"]["level"]
        assert h.window_size == vec["header"]["window_size"]
        return
    r, out = oracle.decompress(FMT[vec["format"]], data, vec["all_members"], cap=4096)
    if "error" in vec:
        assert r.status not in (0, 1), "must be a decompression-error that is not eof"
        return
    assert r.status == 0, oracle_status(r)
    assert out == bytes(vec["output"])
    if "consumed" in vec:
        assert r.in_end_bit == 8 * vec["consumed"]
    hdr = vec.get("header")
    if hdr and vec["format"] == "zlib":
        _, _, h = oracle.zlib_member(data, False, None, 64)
        assert (h.window_size, h.level, h.has_dict) == (hdr["window_size"], hdr["level"], 0)
    if hdr and vec["format"] == "gzip":
        _, _, h = oracle.gzip_member(data, 64)
        assert data[h.name_off:h.name_off + h.name_len].decode("latin-1") == hdr["filename"]
        assert (h.textp, h.mtime, h.extra_flags, h.os) == (0, 0, 0, 0)
        assert h.extra_off == -1 and h.comment_off == -1


def oracle_status(r):
    return f"status {r.status} detail {r.detail}"


def test_trailing_data_status():
    # example.lisp:29-30 and 67-69 are "Trailing data detected." (interface.lisp:125-130)
    z = bytes(GOLD["vectors"][0]["input"]) + b"\x01\x02\x03"
    r, _ = oracle.decompress(1, z, True)
    assert r.status == 170
    r, _ = oracle.decompress(0, bytes([99, 84, 123, 7, 0, 1, 2, 3]), True)
    assert r.status == 170


def test_check_values():
    c = GOLD["checks"]
    m = bytes(c["message"])
    assert oracle.crc32(m) == c["crc32"]
    assert oracle.crc64(m) == c["crc64"]
    assert oracle.bzcrc(m) == c["bzcrc"]
    assert oracle.adler32(m) == c["adler32"]
    assert oracle.sha256(m).hex() == c["sha256"]
    assert oracle.adler32(bytes([1, 33, 7])) == 5177386           # example.lisp:74
    assert oracle.crc32(b"") == 0 and oracle.adler32(b"") == 1


def test_checksums_vs_cpython():
    rng = random.Random(5)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 5551, 5552, 5553, 70000, 300001):
        b = rng.randbytes(n)
        assert oracle.crc32(b) == zlib.crc32(b)
        assert oracle.adler32(b) == zlib.adler32(b)
        assert oracle.sha256(b) == hashlib.sha256(b).digest()
    # Adler-32 worst case for the deferred modulo (adler32.lisp:30-52)
    b = b"\xff" * 200000
    assert oracle.adler32(b) == zlib.adler32(b)


def _deflate(fmt, d, level, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15):
    w = {0: -wbits, 1: wbits, 2: 16 + wbits}[fmt]
    co = zlib.compressobj(level, zlib.DEFLATED, w, 9, strategy)
    return co.compress(d) + co.flush()


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_deflate_family_vs_zlib(fmt):
    rng = random.Random(fmt)
    gens = [corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed]
    for i in range(40):
        n = rng.choice([0, 1, 2, 300, 4096, 40000, 262144])
        d = gens[i % 4](n, seed=i) if n else b""
        blob = _deflate(fmt, d, rng.choice([0, 1, 6, 9]),
                        rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]),
                        rng.choice([9, 12, 15]) if fmt != 2 else 15)
        r, out = oracle.decompress(fmt, blob, True, cap=n + 8)
        assert r.status == 0, (fmt, i, oracle_status(r))
        assert out == d and r.in_end_bit == 8 * len(blob)


def test_gzip_multi_member_and_header_fields():
    parts = [corpus.text(3000, seed=1), b"", corpus.random_bytes(500, seed=2)]
    data = b"".join(_deflate(2, p, 6) for p in parts)
    r, out = oracle.decompress(2, data, True)
    assert r.status == 0 and out == b"".join(parts)
    r, out = oracle.decompress(2, data, False)                    # first member only
    assert r.status == 0 and out == parts[0] and r.in_end_bit == 8 * len(_deflate(2, parts[0], 6))
    r, _ = oracle.decompress(2, data + b"\x00", True)             # garbage after the last member
    assert r.status not in (0,)


def test_deflate_reject_paths():
    # reserved BTYPE (deflate.lisp:397)
    r, _ = oracle.inflate(bytes([0b111]))
    assert r.status == 20
    # stored block LEN/NLEN mismatch (deflate.lisp:383-385)
    r, _ = oracle.inflate(bytes([1, 5, 0, 0, 0]))
    assert r.status == 21
    # truncated input is `eof`, not a data error (common.lisp:56-57)
    blob = _deflate(0, corpus.text(5000, seed=3), 6)
    for cut in (0, 1, len(blob) // 2, len(blob) - 1):
        r, _ = oracle.inflate(blob[:cut], cap=6000)
        assert r.status == 1, cut
    # window enforcement (README.org:184): distance beyond the declared window
    d = corpus.text(3000, seed=3) + corpus.random_bytes(3000, seed=4) + corpus.text(3000, seed=3)
    raw = _deflate(0, d, 9)
    r, o = oracle.inflate(raw, window=32768, cap=len(d))
    assert r.status == 0 and o == d
    r, _ = oracle.inflate(raw, window=1024, cap=len(d))
    assert r.status == 27
    # a single length-1 literal/length code is an underfull tree here (huffman.lisp:166-184; stricter than zlib)
    bits = "1" + "01" + "00000" + "00000" + "0000"          # BFINAL, dynamic, HLIT=257, HDIST=1, HCLEN=4
    bits += "000" * 4                                       # all code-length code lengths zero -> underfull
    by = bytes(int(bits[i:i + 8][::-1].ljust(8, "0")[::-1], 2) if False else
               sum(int(b) << k for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))
    r, _ = oracle.inflate(by + b"\0" * 8)
    assert r.status == 10


def test_zlib_and_gzip_trailer_checks():
    d = corpus.text(2000, seed=8)
    z = bytearray(_deflate(1, d, 6)); z[-1] ^= 1
    r, _ = oracle.decompress(1, bytes(z), True, cap=4000)
    assert r.status == 45
    g = bytearray(_deflate(2, d, 6)); g[-8] ^= 1
    r, _ = oracle.decompress(2, bytes(g), True, cap=4000)
    assert r.status == 55
    g = bytearray(_deflate(2, d, 6)); g[-1] ^= 1
    r, _ = oracle.decompress(2, bytes(g), True, cap=4000)
    assert r.status == 56


@pytest.mark.parametrize("level", [1, 5, 9])
def test_bzip2_vs_cpython(level):
    rng = random.Random(level)
    for i, gen in enumerate([corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed]):
        n = rng.choice([1, 1000, 150000, 400000])
        d = gen(n, seed=10 * level + i)
        blob = bz2.compress(d, level)
        r, out = oracle.decompress(3, blob, True, cap=n + 8)
        assert r.status == 0, oracle_status(r)
        assert out == d and r.in_end_bit == 8 * len(blob)
    # empty stream (no blocks) and concatenated streams (bzip2.lisp:453-454)
    r, out = oracle.decompress(3, bz2.compress(b"", level), True)
    assert r.status == 0 and out == b""
    a, b = corpus.text(5000, seed=1), corpus.mixed(7000, seed=2)
    r, out = oracle.decompress(3, bz2.compress(a, level) + bz2.compress(b, 10 - level), True, cap=20000)
    assert r.status == 0 and out == a + b


def test_bzip2_reject_paths():
    d = corpus.text(20000, seed=4)
    blob = bytearray(bz2.compress(d, 1))
    r, _ = oracle.decompress(3, bytes(blob[:-1]), True, cap=30000)
    assert r.status == 1                                           # eof
    bad = bytearray(blob); bad[-1 - 2] ^= 0x10                     # stream CRC
    r, _ = oracle.decompress(3, bytes(bad), True, cap=30000)
    assert r.status in (74, 63)
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x04              # payload bit flip -> some data error
    r, _ = oracle.decompress(3, bytes(bad), True, cap=30000)
    assert r.status != 0
    r, _ = oracle.decompress(3, b"BZh0" + bytes(blob[4:]), True)
    assert r.status == 62
    r, _ = oracle.decompress(3, b"BZx9" + bytes(blob[4:]), True)
    assert r.status == 61
    r, _ = oracle.decompress(3, b"BAh9" + bytes(blob[4:]), True)
    assert r.status == 60


def _xz(d, check=lzma.CHECK_CRC64, filters=None, block=None):
    if filters is None:
        filters = [{"id": lzma.FILTER_LZMA2, "preset": 6}]
    return lzma.compress(d, lzma.FORMAT_XZ, check, None, filters)


@pytest.mark.parametrize("check", [lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256])
def test_xz_vs_cpython(check):
    rng = random.Random(check)
    for i, gen in enumerate([corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed, corpus.x86ish]):
        n = rng.choice([0, 1, 3000, 70000, 300000])
        d = gen(n, seed=7 * i + check) if n else b""
        for filters in ([{"id": lzma.FILTER_LZMA2, "preset": rng.choice([0, 3, 6, 9])}],
                        [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 6}],
                        [{"id": lzma.FILTER_DELTA, "dist": rng.choice([1, 4, 256])},
                         {"id": lzma.FILTER_LZMA2, "preset": 6, "lc": 1, "lp": 2, "pb": 1}]):
            blob = _xz(d, check, filters)
            r, out = oracle.decompress(8, blob, True, cap=n + 8)
            assert r.status == 0, (i, filters, oracle_status(r))
            assert out == d and r.in_end_bit == 8 * len(blob)


def test_xz_multi_stream_and_padding():
    a, b = corpus.text(4000, seed=1), corpus.x86ish(9000, seed=2)
    data = _xz(a) + b"\0" * 8 + _xz(b, lzma.CHECK_SHA256) + b"\0" * 4
    r, out = oracle.decompress(8, data, True, cap=20000)
    assert r.status == 0 and out == a + b
    r, _ = oracle.decompress(8, _xz(a) + b"\0" * 3, True, cap=20000)      # padding not a multiple of 4 (xz.lisp:134)
    assert r.status in (126, 1)
    r, out = oracle.decompress(8, data, False, cap=20000)
    assert r.status == 0 and out == a


def test_xz_reject_paths():
    d = corpus.text(5000, seed=3)
    blob = bytearray(_xz(d))
    bad = bytearray(blob); bad[7] = 0x02                            # unsupported check type + header CRC
    r, _ = oracle.decompress(8, bytes(bad), True, cap=8000)
    assert r.status in (122, 124)
    bad = bytearray(blob); bad[-1] = 0x5B
    r, _ = oracle.decompress(8, bytes(bad), True, cap=8000)
    assert r.status == 136
    bad = bytearray(blob); bad[40] ^= 0x20                          # payload corruption
    r, _ = oracle.decompress(8, bytes(bad), True, cap=8000)
    assert r.status != 0
    r, _ = oracle.decompress(8, bytes(blob[:-5]), True, cap=8000)
    assert r.status == 1


def test_bcj_x86_and_delta_vs_liblzma():
    # liblzma's raw encoder applies the forward filters; the oracle's inverse must undo them
    d = corpus.x86ish(50000, seed=5)
    for start in (0,):
        enc = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_X86, "start_offset": start},
                                                           {"id": lzma.FILTER_LZMA2, "preset": 0}])
        mid = lzma.decompress(enc, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 0}])
        assert mid != d                                             # the filter did change something
        assert oracle.bcj_x86(mid, start) == d
    for dist in (1, 2, 7, 256):
        enc = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_DELTA, "dist": dist},
                                                           {"id": lzma.FILTER_LZMA2, "preset": 0}])
        mid = lzma.decompress(enc, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 0}])
        assert oracle.delta(mid, dist) == d


BCJ_IDS = {"powerpc": (lzma.FILTER_POWERPC, 5), "ia64": (lzma.FILTER_IA64, 6), "arm": (lzma.FILTER_ARM, 7),
           "armthumb": (lzma.FILTER_ARMTHUMB, 8), "sparc": (lzma.FILTER_SPARC, 9)}


@pytest.mark.parametrize("name", sorted(BCJ_IDS))
def test_other_bcj_filters_vs_liblzma(name):
    # xz-bcj-filters.lisp:45-167: liblzma's raw encoder applies the forward filter, the oracle's inverse undoes it,
    # and the oracle's member decoder accepts the same chain inside a container
    lz_id, xz_id = BCJ_IDS[name]
    d = corpus.random_bytes(120000, seed=9) + corpus.x86ish(30003, seed=2)       # random words hit every pattern
    for start in (0, 4096):
        enc = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lz_id, "start_offset": start}, {"id": lzma.FILTER_LZMA2, "preset": 0}])
        mid = lzma.decompress(enc, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 0}])
        assert mid != d
        assert oracle.bcj_simple(mid, xz_id, start) == d
        blob = xzutil.xz_py(d, lzma.CHECK_CRC32, [{"id": lz_id, "start_offset": start}, {"id": lzma.FILTER_LZMA2, "preset": 1}])
        r, o = oracle.decompress(8, blob, True, cap=len(d) + 16)
        assert r.status == 0 and o == d


def test_arm64_bcj_round_trip():
    # no ARM64 encoder in liblzma 5.2.5: the forward filter is tests/xzutil.py (after XZ Utils 5.4 arm64.c); parity of
    # the decoder direction is pinned by reading xz-bcj-filters.lisp:55-80
    rng = random.Random(3)
    words = []
    for _ in range(30000):
        k = rng.random()
        if k < 0.2:
            words.append(0x94000000 | rng.getrandbits(26))                       # BL
        elif k < 0.4:
            words.append(0x90000000 | (rng.getrandbits(2) << 29) | (rng.getrandbits(19) << 5) | rng.getrandbits(5))  # ADRP
        else:
            words.append(rng.getrandbits(32))
    d = b"".join(w.to_bytes(4, "little") for w in words) + b"\x01\x02\x03"
    for start in (0, 1 << 20):
        enc = xzutil.arm64_encode(d, start)
        assert enc != d
        assert oracle.bcj_simple(enc, 0x0A, start) == d
    blob = xzutil.xz_single_block(xzutil.arm64_encode(d), d, 0x0A, oracle.crc64)
    r, o = oracle.decompress(8, blob, True, cap=len(d) + 16)
    assert r.status == 0 and o == d


def test_lzma2_formats():
    """:raw-lzma2 / :lzma2 through the decompress mirror (lzma2.lisp:205-227) against liblzma's raw encoder."""
    d = corpus.mixed(50000, seed=21)
    raw = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 4, "dict_size": 1 << 16}])
    r, out = oracle.decompress(6, raw, True, opt=1 << 16, cap=len(d) + 8)
    assert r.status == 0 and out == d and r.in_end_bit == 8 * len(raw)
    r, _ = oracle.decompress(6, raw + b"x", True, opt=1 << 16, cap=len(d) + 8)
    assert r.status == 170                                        # trailing data: no multi-member support
    r, out = oracle.decompress(6, raw + b"x", False, opt=1 << 16, cap=len(d) + 8)
    assert r.status == 0 and out == d
    for o, want in ((0, 4096), (1, 6144), (10, 1 << 17), (11, 3 << 16), (39, 3 << 30), (40, 0xFFFFFFFF)):
        r, out = oracle.decompress(7, bytes([o]) + raw, True, cap=len(d) + 8)
        assert (r.status == 0 and out == d) == (want >= 1 << 16), (o, r.status)
        if r.status == 0:
            assert r.in_end_bit == 8 * (len(raw) + 1)
    r, _ = oracle.decompress(7, bytes([41]) + raw, True, cap=len(d) + 8)
    assert r.status == 101 and r.detail == 41                     # "Unrecognized LZMA2 dictionary size byte."
    r, _ = oracle.decompress(7, b"", True)
    assert r.status == 1
