"""GPU parity: bzip2 blocks through the C ABI vs the CPU oracle (bit-exact)."""
import bz2
import random

import numpy as np
import pytest

import corpus
from decompress_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu

BLOCK_MAGIC = 0x314159265359
FOOTER_MAGIC = 0x177245385090


def find_magic(data, magic):
    """Bit offsets of a 48-bit magic (test-side scan; the product has its own in split.cpp)."""
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
    pat = np.unpackbits(np.frombuffer(magic.to_bytes(6, "big"), dtype=np.uint8))
    if bits.size < 48:
        return []
    acc = np.ones(bits.size - 47, dtype=bool)
    for k in range(48):
        acc &= bits[k:bits.size - 47 + k] == pat[k]
    return [int(p) for p in np.nonzero(acc)[0]]


def block_units(data, level, auto=True, caps=None):
    units = []
    for i, p in enumerate(find_magic(data, BLOCK_MAGIC)):
        units.append(_lib.Unit(in_off=p // 8, in_bit_off=p % 8, in_limit=len(data),
                               out_off=_lib.OUT_AUTO if auto else caps[i][0],
                               out_cap=0 if auto else caps[i][1],
                               kind=_lib.UNIT_BZIP2_BLOCK, param0=level))
    return units


def check_stream(engine, data, level, expect=None, cap=None):
    units = block_units(data, level)
    assert units
    cap = cap if cap is not None else (len(expect) if expect is not None else 1 << 22)
    out, res = engine.decode_batch(data, units, cap)
    total = b""
    for u, r in zip(units, res):
        er, eo = oracle.bzip2_block(data, 8 * u.in_off + u.in_bit_off, level, cap=max(cap, 1))
        assert r.status == er.status, (r.status, er.status, r.detail, er.detail)
        if er.status == 0:
            assert r.in_end_bit == er.in_end_bit
            assert out.raw[r.out_off:r.out_off + r.out_len] == eo
            assert bytes(r.check[:4]) == bytes(er.check[:4])
            total += eo
        else:
            assert r.detail == er.detail
    if expect is not None:
        assert total == expect
    return res


@pytest.mark.parametrize("level", [1, 9])
def test_valid_blocks(engine, level):
    random.seed(level)
    for i, n in enumerate([1, 5, 300, 5000, 100000, 250000]):
        for gen in (corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed):
            d = gen(n, seed=100 * i + level)
            check_stream(engine, bz2.compress(d, level), level, d)


def test_full_size_blocks(engine):
    d = corpus.text(2_000_000, seed=77)
    check_stream(engine, bz2.compress(d, 9), 9, d)
    z = bytes(3_000_000)                      # periodic block contents: short inverse-BWT cycles
    check_stream(engine, bz2.compress(z, 9), 9, z)
    r = corpus.repetitive(1_500_000, seed=5)  # long RLE1 runs
    check_stream(engine, bz2.compress(r, 9), 9, r)


def test_multistream_batch(engine):
    parts = [corpus.text(120000, seed=1), corpus.mixed(40000, seed=2), corpus.random_bytes(7000, seed=3)]
    data = b"".join(bz2.compress(p, 1) for p in parts)
    check_stream(engine, data, 1, b"".join(parts))


def test_corrupt_blocks(engine):
    random.seed(11)
    base = bz2.compress(corpus.text(60000, seed=9), 1)
    for t in range(60):
        b = bytearray(base)
        for _ in range(random.choice([1, 1, 3])):
            pos = random.randrange(10, len(b))          # keep the stream header and first magic
            b[pos] ^= 1 << random.randrange(8)
        data = bytes(b)
        units = [u for u in block_units(data, 1)][:1]
        out, res = engine.decode_batch(data, units, 1 << 20)
        er, eo = oracle.bzip2_block(data, 8 * units[0].in_off + units[0].in_bit_off, 1, cap=1 << 20)
        assert res[0].status == er.status, (t, res[0].status, er.status)
        if er.status == 0:
            assert out.raw[:res[0].out_len] == eo
        else:
            assert res[0].detail == er.detail, (t, res[0].status, res[0].detail, er.detail)


def test_truncated_block(engine):
    base = bz2.compress(corpus.text(30000, seed=4), 1)
    for cut in (len(base) - 11, len(base) // 2, 40, 12, 9):
        data = base[:cut]
        u = _lib.Unit(in_off=4, in_limit=len(data), out_off=_lib.OUT_AUTO, kind=_lib.UNIT_BZIP2_BLOCK, param0=1)
        _, res = engine.decode_batch(data, [u], 1 << 20)
        er, _ = oracle.bzip2_block(data, 32, 1, cap=1 << 20)
        assert res[0].status == er.status == 1, (cut, res[0].status, er.status)


def test_faults_in_the_selector_section(engine):
    """The selectors of a full block (about 18000, some 5 KB behind the block header) are taken 2048 bits at a time by
    the whole wave; a piece that holds a run of ones that is no tree, or that reaches the end of the input, goes back to
    the loop that takes them one by one.  Single bit flips and cuts all over that section of a 900 k block, each against
    the oracle: status and detail, and the bytes where the block still decodes."""
    random.seed(23)
    d = corpus.text(900_000, seed=23)
    base = bz2.compress(d, 9)
    u = _lib.Unit(in_off=4, in_limit=len(base), out_off=_lib.OUT_AUTO, kind=_lib.UNIT_BZIP2_BLOCK, param0=9)
    seen = set()
    for t in range(48):
        b = bytearray(base)
        pos = random.randrange(60, 5200)
        b[pos] ^= 1 << random.randrange(8)
        data = bytes(b)
        out, res = engine.decode_batch(data, [u], 1 << 20)
        er, eo = oracle.bzip2_block(data, 32, 9, cap=1 << 20)
        assert res[0].status == er.status, (t, pos, res[0].status, er.status)
        seen.add(er.status)
        if er.status == 0:
            assert out.raw[:res[0].out_len] == eo
        else:
            assert res[0].detail == er.detail, (t, pos, res[0].status, res[0].detail, er.detail)
    assert len(seen) >= 2, seen                                   # (not all the same failure)
    for cut in (80, 300, 311, 1000, 2500, 4000, 5000, 5500, 7000):
        data = base[:cut]
        uu = _lib.Unit(in_off=4, in_limit=len(data), out_off=_lib.OUT_AUTO, kind=_lib.UNIT_BZIP2_BLOCK, param0=9)
        _, res = engine.decode_batch(data, [uu], 1 << 20)
        er, _ = oracle.bzip2_block(data, 32, 9, cap=1 << 20)
        assert res[0].status == er.status, (cut, res[0].status, er.status)
        assert res[0].detail == er.detail, (cut, res[0].detail, er.detail)


def test_out_cap_and_explicit_offsets(engine):
    d = corpus.text(250000, seed=6)
    data = bz2.compress(d, 1)
    n = len(find_magic(data, BLOCK_MAGIC))
    # explicit spans with slack
    units = block_units(data, 1, auto=False, caps=[(i * 120000, 110000) for i in range(n)])
    out, res = engine.decode_batch(data, units, n * 120000)
    got = b"".join(out.raw[r.out_off:r.out_off + r.out_len] for r in res)
    assert all(r.status == 0 for r in res) and got == d
    # capacity too small for the batch: the tail units report SEMZ_E_OUT_CAP
    units = block_units(data, 1)
    _, res = engine.decode_batch(data, units, 150000)
    assert res[0].status == 0 and res[-1].status == 200


def test_narrow_root_tables(engine):
    """Batches of more than 1024 blocks run bz_decode_kernel with 8-bit root tables (more blocks per CU); the option
    forces that variant for the same checks, codes of up to 20 bits included, and one real batch goes the automatic way."""
    engine.set_option("bzip2-tables", 1)
    try:
        for level in (1, 9):
            for i, n in enumerate([1, 300, 5000, 250000]):
                for gen in (corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed):
                    d = gen(n, seed=100 * i + level)
                    check_stream(engine, bz2.compress(d, level), level, d)
        # a steep distribution: a few frequent bytes and many rare ones give code lengths up to the limit
        rnd = random.Random(5)
        steep = bytes(rnd.choices(range(256), weights=[2.0 ** -(k / 9.0) for k in range(256)], k=400000))
        check_stream(engine, bz2.compress(steep, 9), 9, steep)
        test_corrupt_blocks(engine)
        test_truncated_block(engine)
    finally:
        engine.set_option("bzip2-tables", 0)
    parts = [corpus.mixed(1500 + 37 * (i % 41), seed=i) for i in range(1100)]
    data = b"".join(bz2.compress(p, 1) for p in parts)
    check_stream(engine, data, 1, b"".join(parts), cap=1 << 22)


def test_batch_in_parts(engine):
    """A large batch goes in parts, the decode of one beside the inverse BWT of the one before it on a second stream
    (semz_run_bzip2); the option makes parts of 16 blocks so that a batch of this size takes that road, four parts."""
    parts = [corpus.mixed(20000 + 977 * (i % 23), seed=300 + i) for i in range(70)] + [corpus.repetitive(150000, seed=9)]
    data = b"".join(bz2.compress(p, 1) for p in parts)
    engine.set_option("bzip2-overlap-blocks", 16)
    try:
        for _ in range(2):                                        # (twice: the second call meets the scratch of the first)
            check_stream(engine, data, 1, b"".join(parts), cap=1 << 22)
    finally:
        engine.set_option("bzip2-overlap-blocks", 2048)


def test_origptr_on_a_marker_position(engine):
    """Blocks of 17 to 40 KB have walk markers every 2 or 4 positions, so origPtr is a marker position in every second or
    fourth of them, and chains of four times that length go on under ids handed out at run time: the marker's own
    chain would be a twin of origPtr's chain that nothing links to, and when a row it went on in happened to head a
    run, bz_link_kernel claimed the rows behind it a second time (one block of a 4800-block batch failed its CRC in
    every second run).  The twin is not walked any more; 1500 such blocks, three times."""
    parts = [corpus.mixed(17000 + 15 * i, seed=5000 + i) for i in range(1500)]
    blobs = [bz2.compress(p, 1) for p in parts]
    data = b"".join(blobs)
    units, off = [], 0
    for b in blobs:                                               # one block per member, right behind "BZh1"
        units.append(_lib.Unit(in_off=off + 4, in_bit_off=0, in_limit=len(data), out_off=_lib.OUT_AUTO, out_cap=0,
                               kind=_lib.UNIT_BZIP2_BLOCK, param0=1))
        off += len(b)
    want = b"".join(parts)
    for _ in range(3):
        out, res = engine.decode_batch(data, units, len(want))
        assert [r.status for r in res] == [0] * len(units)
        assert out.raw[:len(want)] == want


def test_default_road_beyond_4096_blocks(engine):
    """The road a batch of BASELINE configs[2]'s size takes by itself: more than 4096 blocks, so narrow root tables
    chosen automatically and two parts on two streams (the decode of one part beside the inverse BWT of the other),
    after a stream call whose download has created the context's copy streams -- with more streams than hardware
    queues the second stream of semz_run_bzip2, then created on first use at the highest priority, never saw the
    first one's events and the 4800-block bench hung (round 3).  66 streams of 64 blocks of 100 k, replicated from
    four distinct ones; every status, every block's size and CRC against the streams' own, the first and the last
    stream's bytes."""
    import zlib
    big = corpus.text(40 << 20, seed=4096)                        # a download of 32 MiB and more goes in four slices
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    gz = co.compress(big) + co.flush()
    got, r, _ = engine.decode_stream(_lib.FMT_GZIP, gz, True)
    assert r.status == 0 and got == big
    del got, big
    plains = [corpus.text(6_400_000 - 1000 * i, seed=9000 + i) for i in range(4)]
    streams = [bz2.compress(p, 1) for p in plains]
    per = [block_units(s, 1) for s in streams]
    order = [i % 4 for i in range(66)]
    data = b"".join(streams[i] for i in order)
    units, off = [], 0
    for i in order:
        for u in per[i]:
            units.append(_lib.Unit(in_off=off + u.in_off, in_bit_off=u.in_bit_off, in_limit=len(data), out_off=_lib.OUT_AUTO,
                                   out_cap=0, kind=_lib.UNIT_BZIP2_BLOCK, param0=1))
        off += len(streams[i])
    assert len(units) > 4096
    total = sum(len(plains[i]) for i in order)
    for _ in range(2):
        out, res = engine.decode_batch(data, units, total)
        assert [r.status for r in res] == [0] * len(units)
        assert sum(r.out_len for r in res) == total
        assert out.raw[:len(plains[0])] == plains[0]
        assert out.raw[total - len(plains[order[-1]]):total] == plains[order[-1]]
        # every copy of a stream decodes to the same blocks
        k = 0
        first = {}
        for i in order:
            for j in range(len(per[i])):
                sig = (res[k].out_len, bytes(res[k].check[:4]))
                assert first.setdefault((i, j), sig) == sig
                k += 1
