"""GPU parity: xz blocks / raw LZMA2 through the C ABI vs the CPU oracle (bit-exact)."""
import lzma
import random

import pytest

import corpus
import xzutil
from decompress_amd import _lib
from oracle import oracle
from test_split import split

pytestmark = pytest.mark.gpu


def decode_units(engine, data, units, cap):
    out, res = engine.decode_batch(data, units, cap)
    return out, res


def check_xz(engine, data, expect=None):
    """Splits with the product splitter, decodes the blocks as one batch, compares with the oracle's
    member-level result (the blocks of a valid member concatenate to its output)."""
    units, info = split(_lib.FMT_XZ, data)
    er, eo = oracle.decompress(8, data, True, cap=(len(expect) if expect is not None else 1 << 22) + 16)
    assert units, "splitter found no blocks"
    cap = sum(u.out_cap for u in units)
    out, res = decode_units(engine, data, units, cap + 16)
    got = b"".join(out.raw[r.out_off:r.out_off + r.out_len] for r in res)
    if er.status == 0:
        assert all(r.status == 0 for r in res), [(r.status, r.detail) for r in res]
        assert got == eo
        if expect is not None:
            assert got == expect
    else:
        bad = [r for r in res if r.status != 0]
        assert bad and (bad[0].status, bad[0].detail) == (er.status, er.detail), \
            ([(r.status, r.detail) for r in res], er.status, er.detail)
    return res


@pytest.mark.parametrize("check", [lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256])
def test_single_block_members(engine, check):
    rng = random.Random(check)
    for i, gen in enumerate([corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed, corpus.x86ish]):
        n = rng.choice([1, 3000, 70000, 300000])
        d = gen(n, seed=7 * i + check)
        for filters in ([{"id": lzma.FILTER_LZMA2, "preset": rng.choice([0, 3, 6, 9])}],
                        [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 6}],
                        [{"id": lzma.FILTER_DELTA, "dist": rng.choice([1, 4, 256])},
                         {"id": lzma.FILTER_LZMA2, "preset": 6, "lc": 1, "lp": 2, "pb": 1}],
                        [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_DELTA, "dist": 3},
                         {"id": lzma.FILTER_LZMA2, "preset": 2, "lc": 0, "lp": 4, "pb": 4}]):
            res = check_xz(engine, xzutil.xz_py(d, check, filters), d)
            want_len = {lzma.CHECK_NONE: 0, lzma.CHECK_CRC32: 4, lzma.CHECK_CRC64: 8, lzma.CHECK_SHA256: 32}[check]
            assert res[0].check_len == want_len


def test_multi_block_x86_crc64_and_sha256(engine):
    d = corpus.x86ish(1_500_000, seed=5)
    for chk in ("crc64", "sha256"):
        data = xzutil.xz_cli(d, "--block-size=262144", "--x86", "--lzma2=preset=6", f"--check={chk}")
        res = check_xz(engine, data, d)
        assert len(res) == 6


def test_one_mib_blocks(engine):
    d = corpus.text(3 << 20, seed=6)
    data = xzutil.xz_cli(d, "--block-size=1MiB", "--lzma2=preset=6", "--check=crc64")
    before = engine.get_option("lzma-handed-back")
    res = check_xz(engine, data, d)
    assert len(res) == 3 and all(r.out_len == 1 << 20 for r in res)
    assert engine.get_option("lzma-handed-back") == before, "a well-formed block took the general kernel"


def test_well_formed_blocks_stay_on_the_fast_kernel(engine):
    """The accept-only kernel (lzma2_fast.hip) decodes every kind of well-formed block with ordinary properties by
    itself: text, x86 code behind the BCJ filter, incompressible data (stored chunks), long runs (matches closer than
    they are long, and the longest ones), every preset's match finder.  What it hands back is decoded correctly by
    the general kernel all the same, so only this counter shows a fast path that has stopped accepting."""
    before = engine.get_option("lzma-handed-back")
    for i, gen in enumerate([corpus.text, corpus.x86ish, corpus.random_bytes, corpus.repetitive, corpus.mixed]):
        d = gen(700_000, seed=90 + i)
        for preset in (0, 6, 9):
            check_xz(engine, xzutil.xz_py(d, lzma.CHECK_CRC64, [{"id": lzma.FILTER_LZMA2, "preset": preset}]), d)
        check_xz(engine, xzutil.xz_cli(d, "--block-size=262144", "--x86", "--lzma2=preset=6", "--check=crc64"), d)
    assert engine.get_option("lzma-handed-back") == before
    # the counter counts: lc + lp = 4 is beyond the fast kernel's literal table
    d = corpus.text(50_000, seed=99)
    check_xz(engine, xzutil.xz_py(d, lzma.CHECK_CRC32, [{"id": lzma.FILTER_LZMA2, "preset": 6, "lc": 2, "lp": 2}]), d)
    assert engine.get_option("lzma-handed-back") == before + 1


def test_config4_shape_one_mib_blocks_x86_sha256(engine):
    """BASELINE configs[3] as stated: 1 MiB LZMA2 blocks behind the x86 BCJ filter with SHA-256 (and CRC64) checks,
    every block's check verified on the device; also through the stream entry point (index, footer)."""
    d = corpus.x86ish(4 * (1 << 20) + 12345, seed=31)
    for chk, clen in (("sha256", 32), ("crc64", 8)):
        data = xzutil.xz_cli(d, "--block-size=1MiB", "--x86", "--lzma2=preset=6", f"--check={chk}")
        res = check_xz(engine, data, d)
        assert len(res) == 5 and all(r.out_len == 1 << 20 for r in res[:4]) and all(r.check_len == clen for r in res)
        if chk == "sha256":
            import hashlib
            for k, r in enumerate(res):
                assert bytes(r.check[:32]) == hashlib.sha256(d[k << 20:(k + 1) << 20]).digest()
        got, sres, info = engine.decode_stream(_lib.FMT_XZ, data, True)
        assert sres.status == 0 and got == d and info.first_header[0] == (10 if chk == "sha256" else 4)
        # a damaged block check is found on the device and reported with the stored value's first word
        bad = bytearray(data)
        units, _ = split(_lib.FMT_XZ, data)
        pos = units[2].in_off - 16 - 1                 # last byte of block 1's check (block 2's header: 12 bytes + CRC-32)
        bad[pos] ^= 0x80
        got, sres, _ = engine.decode_stream(_lib.FMT_XZ, bytes(bad), True)
        er, eo = oracle.decompress(8, bytes(bad), True, cap=len(d) + 16)
        assert (sres.status, sres.detail) == (er.status, er.detail) and sres.status == 155 and got == d[:1 << 20]


def test_reference_lol_vector(engine):
    # README.org:306-308: LZMA chunk -> stored chunks -> LZMA chunk without state reset
    import base64
    data = base64.b64decode("/Td6WFoAAAD/EtlBAgAhAQoAAABTxyq54AAAAAUJACX//AAAAgAAT4AAAAAFACfRR0AAAAABKAM7StLkBnKeegEAAAAAAFla")
    check_xz(engine, data, b"LOL")


def test_raw_lzma2_units(engine):
    items = []
    for i, n in enumerate([1, 500, 40000, 200000]):
        d = corpus.mixed(n, seed=i)
        f = [{"id": lzma.FILTER_LZMA2, "preset": 6, "dict_size": 1 << 20}]
        items.append((lzma.compress(d, lzma.FORMAT_RAW, filters=f), d))
    data = b"".join(b for b, _ in items)
    units, off, ooff = [], 0, 0
    for b, d in items:
        units.append(_lib.Unit(in_off=off, in_limit=off + len(b), out_off=ooff, out_cap=len(d), kind=_lib.UNIT_LZMA2,
                               param0=1 << 20))
        off += len(b)
        ooff += len(d)
    out, res = engine.decode_batch(data, units, ooff)
    for u, r, (b, d) in zip(units, res, items):
        er, eo = oracle.lzma2(b, 1 << 20, cap=len(d))
        assert r.status == er.status == 0
        assert out.raw[u.out_off:u.out_off + r.out_len] == eo == d
        assert r.in_end_bit - 8 * u.in_off == er.in_end_bit


def test_corrupt_and_truncated_lzma2(engine):
    rng = random.Random(3)
    d = corpus.mixed(60000, seed=9)
    f = [{"id": lzma.FILTER_LZMA2, "preset": 4, "dict_size": 1 << 16}]
    base = lzma.compress(d, lzma.FORMAT_RAW, filters=f)
    blobs = [base[:cut] for cut in (0, 1, 3, 7, len(base) // 2, len(base) - 1)]
    for _ in range(80):
        b = bytearray(base)
        for _ in range(rng.choice([1, 1, 2, 4])):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        blobs.append(bytes(b))
    # small dictionary: distances beyond it are rejected (lzma.lisp:604-605)
    data = b"".join(blobs)
    units, off = [], 0
    for i, b in enumerate(blobs):
        units.append(_lib.Unit(in_off=off, in_limit=off + len(b), out_off=i * 70000, out_cap=66000, kind=_lib.UNIT_LZMA2,
                               param0=(1 << 16) if i % 3 else 4096))
        off += len(b)
    out, res = engine.decode_batch(data, units, len(blobs) * 70000)
    for u, r, b in zip(units, res, blobs):
        er, eo = oracle.lzma2(b, u.param0, cap=66000)
        assert r.status == er.status, (len(b), r.status, er.status, r.detail, er.detail)
        if er.status == 0:
            assert out.raw[u.out_off:u.out_off + r.out_len] == eo
            assert r.in_end_bit - 8 * u.in_off == er.in_end_bit
        else:
            assert r.detail == er.detail


def test_corrupt_xz_blocks(engine):
    rng = random.Random(5)
    d = corpus.x86ish(50000, seed=2)
    base = xzutil.xz_py(d, lzma.CHECK_CRC32, [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 3}])
    units0, _ = split(_lib.FMT_XZ, base)
    data_lo = units0[0].in_off
    for t in range(40):
        b = bytearray(base)
        for _ in range(rng.choice([1, 2])):
            b[rng.randrange(data_lo, len(b) - 24)] ^= 1 << rng.randrange(8)     # payload, padding or check
        data = bytes(b)
        u = units0[0]
        out, res = engine.decode_batch(data, [u], u.out_cap + 16)
        er, eo = oracle.decompress(8, data, True, cap=len(d) + 16)
        assert res[0].status == er.status, (t, res[0].status, er.status)
        if er.status == 0:
            assert out.raw[:res[0].out_len] == eo
        else:
            assert res[0].detail == er.detail, (t, res[0].status, res[0].detail, er.detail)


def test_x86_bcj_at_any_output_alignment(engine):
    """bcj_x86_kernel takes 16-byte loads and stores when a block's output starts on a 16-byte boundary and the byte-wise
    road otherwise; operands that straddle its 1 KiB pieces are handed on in LDS.  The same blocks at sixteen output
    offsets, sizes around the piece size and its multiples, against the oracle."""
    rng = random.Random(17)
    for n in (5, 900, 1019, 1024, 1029, 2048, 5000, 70001):
        d = corpus.x86ish(n, seed=n)
        data = xzutil.xz_py(d, lzma.CHECK_CRC32, [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 3}])
        units0, _ = split(_lib.FMT_XZ, data)
        er, eo = oracle.decompress(8, data, True, cap=n + 16)
        assert er.status == 0 and eo == d
        assert len(units0) == 1
        for shift in rng.sample(range(64), 16) + [0]:
            u = units0[0]
            u2 = _lib.Unit(in_off=u.in_off, in_bit_off=u.in_bit_off, in_limit=u.in_limit, out_off=shift, out_cap=u.out_cap, kind=u.kind,
                           flags=u.flags, param0=u.param0, param1=u.param1, param2=u.param2, param3=u.param3, filters=u.filters,
                           size_in=u.size_in, size_out=u.size_out)
            out, res = engine.decode_batch(data, [u2], shift + u.out_cap + 16)
            assert res[0].status == 0, (n, shift, res[0].status, res[0].detail)
            assert out.raw[shift:shift + res[0].out_len] == d, (n, shift)


@pytest.mark.parametrize("name", ["powerpc", "ia64", "arm", "armthumb", "sparc"])
def test_other_bcj_filters(engine, name):
    # xz-bcj-filters.lisp:45-167, chains as the reference builds them (xz.lisp:311-344)
    lz_id = {"powerpc": lzma.FILTER_POWERPC, "ia64": lzma.FILTER_IA64, "arm": lzma.FILTER_ARM,
             "armthumb": lzma.FILTER_ARMTHUMB, "sparc": lzma.FILTER_SPARC}[name]
    d = corpus.random_bytes(250001, seed=4) + corpus.x86ish(40000, seed=8)
    for filters in ([{"id": lz_id}, {"id": lzma.FILTER_LZMA2, "preset": 3}],
                    [{"id": lz_id, "start_offset": 1 << 16}, {"id": lzma.FILTER_LZMA2, "preset": 1}],
                    [{"id": lzma.FILTER_DELTA, "dist": 4}, {"id": lz_id}, {"id": lzma.FILTER_LZMA2, "preset": 1}]):
        check_xz(engine, xzutil.xz_py(d, lzma.CHECK_CRC64, filters), d)
    # unaligned block starts inside the output: multi-block stream through the CLI
    data = xzutil.xz_cli(d, "--block-size=70001", f"--{name}", "--lzma2=preset=1", "--check=crc32")
    check_xz(engine, data, d)


def test_arm64_bcj_filter(engine):
    rng = random.Random(5)
    words = [0x94000000 | rng.getrandbits(26) if rng.random() < 0.2 else
             0x90000000 | (rng.getrandbits(2) << 29) | (rng.getrandbits(19) << 5) | rng.getrandbits(5) if rng.random() < 0.25 else
             rng.getrandbits(32) for _ in range(60000)]
    d = b"".join(w.to_bytes(4, "little") for w in words) + b"\x07"
    check_xz(engine, xzutil.xz_single_block(xzutil.arm64_encode(d), d, 0x0A, oracle.crc64), d)
