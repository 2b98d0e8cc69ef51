"""CPU: the host splitter (semz_split) finds the right units without a GPU."""
import bz2
import ctypes as C
import lzma

import corpus
import xzutil
from decompress_amd import _lib


def split(fmt, data, offsets=None, out_sizes=None):
    lib = _lib.load()
    units = C.POINTER(_lib.Unit)()
    n = C.c_size_t()
    info = _lib.StreamInfo()
    no = len(offsets) if offsets else 0
    offs = (C.c_uint64 * no)(*offsets) if no else None
    sizes = (C.c_uint64 * no)(*out_sizes) if out_sizes else None
    rc = lib.semz_split(fmt, data, len(data), offs, sizes, no, C.byref(units), C.byref(n), C.byref(info))
    assert rc == 0
    out = []
    for i in range(n.value):
        u = _lib.Unit()
        C.memmove(C.byref(u), C.byref(units[i]), C.sizeof(_lib.Unit))
        out.append(u)
    lib.semz_free(units)
    return out, info


def test_bgzf_walk():
    d = corpus.text(250000, seed=1)
    data = xzutil.bgzf(d, 60000)
    units, info = split(_lib.FMT_GZIP, data)
    assert len(units) == 5 and info.reserved == 5
    assert [u.out_cap for u in units] == [60000, 60000, 60000, 60000, 10000]
    assert units[0].in_off == 0 and all(u.in_limit == len(data) for u in units)
    assert [u.out_off for u in units] == [0, 60000, 120000, 180000, 240000]
    # a plain gzip member after the indexed ones is left as an unindexed tail
    import gzip
    data2 = data + gzip.compress(b"tail")
    units, info = split(_lib.FMT_GZIP, data2)
    assert len(units) == 6 and info.reserved == 5 and units[5].in_off == len(data)


def test_record_offsets():
    import zlib
    recs = [zlib.compress(corpus.text(4096, seed=i)) for i in range(10)]
    offs, o = [], 0
    for r in recs:
        offs.append(o)
        o += len(r)
    units, info = split(_lib.FMT_ZLIB, b"".join(recs), offs, [4096] * 10)
    assert len(units) == 10 and info.total_out == 40960
    assert [u.in_off for u in units] == offs
    assert [u.in_limit for u in units] == offs[1:] + [o]
    assert all(u.kind == _lib.UNIT_ZLIB and u.out_cap == 4096 for u in units)


def test_bzip2_magic_scan():
    parts = [corpus.text(250000, seed=1), corpus.mixed(120000, seed=2)]
    data = bz2.compress(parts[0], 1) + bz2.compress(parts[1], 3)
    units, info = split(_lib.FMT_BZIP2, data)
    assert info.first_header[0] == 1 and info.nmembers == 2
    assert len(units) == 3 + 1                       # 250000 bytes at level 1 = 3 blocks; 120000 at level 3 = 1
    assert [u.param0 for u in units] == [1, 1, 1, 3]
    assert units[0].in_off == 4 and units[0].in_bit_off == 0
    assert all(u.out_off == _lib.OUT_AUTO and u.kind == _lib.UNIT_BZIP2_BLOCK for u in units)
    # block starts are at arbitrary bit offsets
    assert any(u.in_bit_off for u in units)


def test_xz_index_walk():
    d = corpus.x86ish(300000, seed=3)
    data = xzutil.xz_cli(d, "--block-size=65536", "--x86", "--lzma2=preset=6", "--check=crc64")
    units, info = split(_lib.FMT_XZ, data)
    assert info.first_header[0] == 4
    assert len(units) == 5 and sum(u.out_cap for u in units) == len(d)
    for u in units:
        assert u.kind == _lib.UNIT_XZ_BLOCK and u.filters == _lib.FILTER_X86 and u.param1 == 4
        assert u.param0 == 1 << 23                   # preset 6 dictionary
    # two streams with padding in between, different checks
    data2 = data + b"\0" * 8 + xzutil.xz_py(b"second", lzma.CHECK_SHA256)
    units, _ = split(_lib.FMT_XZ, data2)
    assert len(units) == 6 and units[5].param1 == 10 and units[5].filters == 0
    # delta + x86 chain: the filter next to LZMA2 is applied first
    f = [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_DELTA, "dist": 4}, {"id": lzma.FILTER_LZMA2, "preset": 1}]
    units, _ = split(_lib.FMT_XZ, xzutil.xz_py(d[:5000], filters=f))
    assert units[0].filters == (_lib.FILTER_DELTA | (_lib.FILTER_X86 << 4)) and units[0].param3 == 3
    # a broken footer: nothing is predicted, the forward pass of semz_decode_stream takes over
    bad = bytearray(data); bad[-1] ^= 1
    units, _ = split(_lib.FMT_XZ, bytes(bad))
    assert units == []


def test_bzip2_magic_scan_large_input_in_slices():
    """Inputs of 8 MiB and more are scanned in slices by several host threads: magics at every bit offset, also
    across the slice borders, come out in order and none twice."""
    import random
    rng = random.Random(7)
    n = 9 << 20
    buf = bytearray(rng.getrandbits(8) for _ in range(1 << 16)) * (n >> 16)
    head = b"BZh9"
    buf[:4] = head
    magic = 0x314159265359
    want = []
    borders = [n * t // 8 for t in range(1, 8)]
    # one magic per slice border, each straddling it differently (they must not overlap each other)
    places = [32 + 3, 1000 * 8 + 5] + [8 * b - 48 + d for b, d in zip(borders, (-9, 0, 1, 7, 23, 47, -47))] + [8 * n - 48]
    for bit in sorted(set(places)):
        # write the 48-bit pattern at an arbitrary bit offset, MSB first
        for k in range(48):
            pos = bit + k
            byte, sh = pos >> 3, 7 - (pos & 7)
            b = (magic >> (47 - k)) & 1
            buf[byte] = (buf[byte] & ~(1 << sh)) | (b << sh)
        want.append(bit)
    units, info = split(_lib.FMT_BZIP2, bytes(buf))
    got = [8 * u.in_off + u.in_bit_off for u in units]
    assert got == sorted(got) and len(set(got)) == len(got)
    assert set(want) <= set(got), sorted(set(want) - set(got))
    # anything else the scan reports must really be a magic (random data: practically none)
    for bit in set(got) - set(want):
        v = int.from_bytes(bytes(buf[bit >> 3:(bit >> 3) + 7]), "big") >> (8 - (bit & 7)) & ((1 << 48) - 1)
        assert v == magic
