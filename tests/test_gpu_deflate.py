"""GPU parity: Deflate / zlib / gzip units through the C ABI vs the CPU oracle.

Bar (integer/byte work): bit-exact output, identical status codes and detail
values, identical consumed-bit positions and check values.
"""
import gzip
import random
import zlib

import pytest

import corpus
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu

KIND_OF_FMT = {0: _lib.UNIT_DEFLATE, 1: _lib.UNIT_ZLIB, 2: _lib.UNIT_GZIP}
TRAILING = {0: 0, 1: 32, 2: 64}


def oracle_unit(fmt, blob, cap, window=0, prefix=None, have_dict_fn=False):
    if fmt == 0:
        return oracle.inflate(blob, 0, 0, window or 32768, prefix, cap)
    if fmt == 1:
        r, o, _ = oracle.zlib_member(blob, have_dict_fn, prefix, cap)
        return r, o
    r, o, _ = oracle.gzip_member(blob, cap)
    return r, o


def run_batch(engine, items, seed=None):
    """items: list of (fmt, blob, cap[, window]) decoded as one batch; returns per-item (Result, bytes)."""
    if seed is not None:
        random.seed(seed)
    data = bytearray()
    units = []
    out_off = 0
    for it in items:
        fmt, blob, cap = it[0], it[1], it[2]
        window = it[3] if len(it) > 3 else 0
        # misalign inputs and outputs on purpose
        pad = random.randrange(0, 5)
        data += b"\xAA" * pad
        u = _lib.Unit(in_off=len(data), in_limit=len(data) + len(blob), out_off=out_off, out_cap=cap,
                      kind=KIND_OF_FMT[fmt], param0=window)
        data += blob
        units.append(u)
        out_off += cap + random.randrange(0, 7)
    out, res = engine.decode_batch(bytes(data), units, out_off + 16)
    got = []
    for u, r in zip(units, res):
        got.append((r, out.raw[u.out_off:u.out_off + r.out_len], u))
    return got


def check_items(engine, items, seed=None):
    got = run_batch(engine, items, seed)
    for it, (r, o, u) in zip(items, got):
        fmt, blob, cap = it[0], it[1], it[2]
        window = it[3] if len(it) > 3 else 0
        er, eo = oracle_unit(fmt, blob, cap, window)
        assert r.status == er.status, (fmt, len(blob), r.status, er.status, r.detail, er.detail)
        if er.status == 0:
            assert o == eo
            assert r.in_end_bit - 8 * u.in_off == er.in_end_bit
            assert bytes(r.check[:r.check_len]) == bytes(er.check[:er.check_len])
        else:
            assert r.detail == er.detail, (fmt, r.status, r.detail, er.detail)


def compress(fmt, d, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15):
    if fmt == 2:
        co = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, 9, strategy)
    elif fmt == 1:
        co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, -wbits, 9, strategy)
    return co.compress(d) + co.flush()


@pytest.mark.parametrize("group", [1, 2, 3])
def test_example_vectors(engine, group):
    # example.lisp:6-7, 12-19, 24-26
    z = bytes([120, 156, 99, 84, 100, 7, 0, 0, 79, 0, 42])
    g = bytes([31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 102, 111, 111, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0])
    got = run_batch(engine, [(1, z, 16), (2, g, 16), (0, z[2:7], 16), (0, bytes([99, 84, 123, 7, 0]), 16)], group)
    assert [o for _, o, _ in got] == [b"\x01\x21\x07", b"\x01\x21\x07", b"\x01\x21\x07", b"\x01\x26\xee"]
    assert all(r.status == 0 for r, _, _ in got)
    assert bytes(got[0][0].check[:4]) == (0x004F002A).to_bytes(4, "little")   # Adler-32
    assert bytes(got[1][0].check[:4]) == (0xEC783365).to_bytes(4, "little")   # CRC-32


@pytest.mark.parametrize("group", [64, 32, 16])
def test_valid_streams(engine, group):  # three seeded corpora
    random.seed(100 + group)
    items = []
    gens = [corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed]
    for i in range(96):
        n = random.choice([0, 1, 2, 7, 100, 1000, 4096, 30000, 70000, 262144])
        d = gens[i % 4](n, seed=i) if n else b""
        fmt = i % 3
        level = random.choice([0, 1, 4, 6, 9])
        strat = random.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        items.append((fmt, compress(fmt, d, level, strat), n + random.randrange(0, 3)))
    check_items(engine, items, group)


def test_multiblock_and_sync_flush(engine):
    # several blocks of different types inside one member, empty stored blocks (Z_SYNC_FLUSH)
    d = corpus.mixed(200000, seed=7)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    blob = b""
    for i in range(0, len(d), 17000):
        blob += co.compress(d[i:i + 17000]) + co.flush(zlib.Z_SYNC_FLUSH if (i // 17000) % 2 else zlib.Z_FULL_FLUSH)
    blob += co.flush()
    check_items(engine, [(0, blob, len(d))] * 3, 64)


@pytest.mark.parametrize("group", [64, 16])
def test_truncated_and_corrupt(engine, group):
    random.seed(7 + group)
    items = []
    for i in range(40):
        fmt = i % 3
        d = corpus.text(random.choice([50, 3000, 40000]), seed=1000 + i)
        blob = compress(fmt, d, random.choice([1, 6, 9]),
                        random.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]))
        cut = random.randrange(0, len(blob))
        items.append((fmt, blob[:cut], len(d) + 8))                      # truncation -> eof or data error
    for i in range(120):
        fmt = i % 3
        d = corpus.mixed(random.choice([200, 5000, 33000]), seed=2000 + i)
        blob = bytearray(compress(fmt, d, 6))
        for _ in range(random.choice([1, 1, 2, 5])):
            blob[random.randrange(0, len(blob))] ^= 1 << random.randrange(8)
        items.append((fmt, bytes(blob), len(d) + 600))                   # corruption -> any reject path
    check_items(engine, items, group)


def test_faults_in_the_dynamic_header(engine):
    """The code lengths of a dynamic block are taken in 64-bit windows by the whole wave; a window that holds one of
    the reference's failures (a repeat with nothing before it, a run beyond the last code, an incomplete tree
    afterwards) or comes near the end of the input goes back to the loop that takes them one by one.  Every bit of the
    first 96 bytes of three raw streams flipped in turn, and cuts all over the header, each against the oracle."""
    seen = set()
    for seed, gen, n in ((1, corpus.text, 40000), (2, corpus.mixed, 9000), (3, corpus.repetitive, 30000)):
        d = gen(n, seed=seed)
        blob = compress(0, d, 6)
        items = []
        for byte in range(min(96, len(blob))):
            for bit in range(8):
                b = bytearray(blob)
                b[byte] ^= 1 << bit
                items.append((0, bytes(b), len(d) + 600))
        for cut in range(2, min(140, len(blob)), 3):
            items.append((0, blob[:cut], len(d) + 8))
        for i in range(0, len(items), 64):
            check_items(engine, items[i:i + 64], 64)
        for it in items[::7]:
            seen.add(oracle_unit(0, it[1], it[2])[0].status)
    assert len(seen) >= 4, seen                                         # (several different failures among them)


def test_more_than_64_tokens_in_a_round(engine):
    """A round of a unit with more than 8 KiB of room holds up to 128 queued tokens, two per lane when they are resolved
    (64 for smaller units).  Short-match text, then bytes that only a stored block takes, then text again, all inside
    2 KiB of output and flushed into blocks of their own: matches and stored spans in either half of the queue, matches
    of the second half that copy from spans and matches of the first.  The same streams with little room take the
    64-token road."""
    items = []
    for seed in range(24):
        rng = random.Random(seed)
        co = zlib.compressobj(9, zlib.DEFLATED, -15)
        blob, plain = b"", b""
        for k in range(rng.choice([3, 5, 8])):
            if k % 2 == 0:
                d = corpus.text(rng.choice([500, 900, 1400]), seed=100 * seed + k)
                if plain and rng.randrange(2):
                    d = d[:200] + plain[-150:] + d[200:]              # copies out of what came before
            else:
                d = corpus.random_bytes(rng.choice([40, 300, 700]), seed=100 * seed + k)
            plain += d
            blob += co.compress(d) + co.flush(rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_BLOCK]))
        blob += co.flush()
        assert zlib.decompress(blob, -15) == plain
        items.append((0, blob, len(plain) + 9000))
        items.append((0, blob, len(plain) + rng.choice([0, 5])))
    check_items(engine, items, 64)


def test_trailer_mismatches(engine):
    d = corpus.text(5000, seed=5)
    g = bytearray(compress(2, d))
    z = bytearray(compress(1, d))
    items = []
    for pos in (-8, -5, -4, -1):
        b = bytearray(g); b[pos] ^= 0x40; items.append((2, bytes(b), len(d)))
    for pos in (-4, -1):
        b = bytearray(z); b[pos] ^= 0x01; items.append((1, bytes(b), len(d)))
    # truncated trailers: CRC present but ISIZE cut; nothing present
    items += [(2, bytes(g[:-3]), len(d)), (2, bytes(g[:-6]), len(d)), (1, bytes(z[:-2]), len(d))]
    # a wrong CRC together with a cut ISIZE: the CRC error wins (gzip.lisp:105-113)
    b = bytearray(g[:-2]); b[-6] ^= 1; items.append((2, bytes(b), len(d)))
    check_items(engine, items)


def test_gzip_header_fields(engine):
    d = corpus.text(300, seed=9)
    body = compress(0, d)
    tail = zlib.crc32(d).to_bytes(4, "little") + len(d).to_bytes(4, "little")
    def member(flg, extra=b"", name=b"", comment=b"", hcrc=None, magic=b"\x1f\x8b", cm=8):
        h = magic + bytes([cm, flg]) + b"\0\0\0\0\0\3"
        if flg & 4: h += len(extra).to_bytes(2, "little") + extra
        if flg & 8: h += name + b"\0"
        if flg & 16: h += comment + b"\0"
        if flg & 2: h += ((zlib.crc32(h) & 0xFFFF) if hcrc is None else hcrc).to_bytes(2, "little")
        return h + body + tail
    items = [
        (2, member(0x1F, b"BC\x02\x00\x12\x34", b"foo", b"a comment"), len(d)),
        (2, member(0x02), len(d)),
        (2, member(0x02, hcrc=0x1234), len(d)),                       # header CRC mismatch
        (2, member(0x04, b"AB\x05\x00xy"), len(d)),                   # inconsistent extra field
        (2, member(0x04, b"AB\x01"), len(d)),                         # no room for the length field
        (2, member(0x20), len(d)),                                    # reserved flag
        (2, member(0, magic=b"\x1f\x8c"), len(d)),
        (2, member(0, cm=7), len(d)),
        (2, member(0x08, name=b"unterminated")[:18], len(d)),         # eof inside the name
        (2, b"\x1f\x8b\x08", len(d)),
    ]
    check_items(engine, items)


def test_zlib_header_checks(engine):
    d = b"hello hello hello"
    z = compress(1, d)
    bad_method = bytes([0x79, z[1]]) + z[2:]
    bad_cinfo = bytes([0x88, 0x1C]) + z[2:]
    bad_fcheck = bytes([z[0], z[1] ^ 1]) + z[2:]
    fdict = bytes([120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42])    # example.lisp:72 without a dictionary fn
    check_items(engine, [(1, bad_method, 64), (1, bad_cinfo, 64), (1, bad_fcheck, 64), (1, fdict, 64),
                         (1, z[:1], 64), (1, b"", 64)])


def test_zlib_window_enforced(engine):
    # README.org:184: a reference that stays in bounds but exceeds the CINFO window is an error
    d = corpus.text(3000, seed=3) + corpus.random_bytes(3000, seed=4) + corpus.text(3000, seed=3)
    z = bytearray(compress(1, d, 9))
    ok = bytes(z)
    z[0] = 0x08 | (0 << 4)                                  # CINFO 0: 256-byte window
    z[1] = (z[1] & 0xE0)
    z[1] |= (31 - ((z[0] * 256 + z[1]) % 31)) % 31
    check_items(engine, [(1, ok, len(d)), (1, bytes(z), len(d))])
    r, _, _ = run_batch(engine, [(1, bytes(z), len(d))])[0]
    assert r.status == 27                                    # SEMZ_E_DEFLATE_DISTANCE
    # raw deflate with an explicit small window (deflate.lisp:412-416)
    raw = compress(0, d, 9)
    check_items(engine, [(0, raw, len(d), 1024), (0, raw, len(d), 32768)])


def test_preset_dictionary(engine):
    # example.lisp:72-74
    blob = bytes([120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42])
    dic = bytes([1, 33, 7])
    data = blob + dic
    u = _lib.Unit(in_off=0, in_limit=len(blob), out_off=0, out_cap=16, kind=_lib.UNIT_ZLIB,
                  flags=_lib.UF_HAVE_DICT_FN, prefix_off=len(blob), prefix_len=len(dic))
    out, res = engine.decode_batch(data, [u], 16)
    assert res[0].status == 0 and out.raw[:res[0].out_len] == b"\x01\x21\x07"
    assert res[0].aux == 5177386
    er, eo, h = oracle.zlib_member(blob, True, dic, 16)
    assert er.status == 0 and eo == b"\x01\x21\x07" and h.dict_id == 5177386
    # longer dictionary-based stream
    dic = corpus.text(20000, seed=11)
    d = dic[5000:9000] + corpus.text(3000, seed=12) + dic[-3000:]
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY, dic)
    blob = co.compress(d) + co.flush()
    data = blob + dic
    u = _lib.Unit(in_off=0, in_limit=len(blob), out_off=3, out_cap=len(d), kind=_lib.UNIT_ZLIB,
                  flags=_lib.UF_HAVE_DICT_FN, prefix_off=len(blob), prefix_len=len(dic))
    out, res = engine.decode_batch(data, [u], len(d) + 8)
    assert res[0].status == 0 and out.raw[3:3 + res[0].out_len] == d
    # dictionary function present but dictionary unknown
    u2 = _lib.Unit(in_off=0, in_limit=len(blob), out_off=0, out_cap=len(d), kind=_lib.UNIT_ZLIB,
                   flags=_lib.UF_HAVE_DICT_FN)
    _, res = engine.decode_batch(data, [u2], len(d) + 8)
    assert res[0].status == 44 and res[0].detail == zlib.adler32(dic)


def test_out_cap(engine):
    d = corpus.text(10000, seed=21)
    items = [(f, compress(f, d), cap) for f in (0, 1, 2) for cap in (0, 1, 9999)]
    got = run_batch(engine, items)
    assert all(r.status == 200 for r, _, _ in got)            # SEMZ_E_OUT_CAP
    check_items(engine, items)


def test_lookahead_into_next_member(engine):
    # all-members mode: in_limit is the end of the input, members are back to back
    parts = [corpus.text(n, seed=30 + i) for i, n in enumerate([10, 5000, 1, 70000, 0, 300])]
    blobs = [compress(2, p, 6) for p in parts]
    data = b"".join(blobs)
    units, off, ooff = [], 0, 0
    for p, b in zip(parts, blobs):
        units.append(_lib.Unit(in_off=off, in_limit=len(data), out_off=ooff, out_cap=len(p), kind=_lib.UNIT_GZIP))
        off += len(b)
        ooff += len(p)
    out, res = engine.decode_batch(data, units, ooff)
    assert out.raw[:ooff] == b"".join(parts)
    off = 0
    for r, b in zip(res, blobs):
        off += len(b)
        assert r.status == 0 and r.in_end_bit == 8 * off
