"""GPU: one input, several devices inside the C call (semz_open_multi).  The box has one GPU, so the device is named
twice or three times: two / three contexts, each with its own stream, arena and host thread, run through exactly
the code a multi-GPU node runs (partition, per-range upload, merge, the reference's walk over the merged results).
The results must be those of the one-device context, byte for byte and status for status."""
import bz2
import gzip
import lzma
import os
import random
import zlib

import pytest

import corpus
from decompress_amd import _lib

pytestmark = pytest.mark.gpu


def _decode(eng, fmt, data, all_members=True):
    d = eng.stream_decode(fmt, data, all_members)
    try:
        res = d.res
        return res.status, res.detail, res.in_end_bit, d.read(0, res.out_len), list(d.members)
    finally:
        d.release()


@pytest.fixture(scope="module")
def engines():
    one = _lib.Engine(0)
    two = _lib.Engine([0, 0])
    three = _lib.Engine([0, 0, 0])
    assert one.device_count == 1 and two.device_count == 2 and three.device_count == 3
    yield one, two, three
    for e in (one, two, three):
        e.close()


def _members(n, size, seed):
    return [corpus.text(size + 977 * (i % 5), seed=seed + i) for i in range(n)]


def test_gzip_members_spread(engines):
    one, two, three = engines
    plain = _members(40, 96 * 1024, 100)
    data = b"".join(gzip.compress(p, 6) for p in plain)
    assert len(data) > (1 << 20)
    want = _decode(one, _lib.FMT_GZIP, data)
    assert want[0] == 0 and want[3] == b"".join(plain)
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_GZIP, data) == want
        # every device, the primary included, is sent the bytes of its own range and a margin: the input goes up once
        assert len(data) <= eng.get_option("uploaded-bytes") <= len(data) + eng.device_count * (64 * 1024 + 64)
    _decode(one, _lib.FMT_GZIP, data)
    assert one.get_option("uploaded-bytes") == len(data)


def test_ranges_follow_the_output_sizes(engines):
    """Members of very different sizes: the ranges are cut by output bytes (ISIZE), SURVEY.md section 8e, so the
    device that gets the large members gets few of them; results are those of one device."""
    one, two, three = engines
    plain = [corpus.text(900_000, seed=400 + i) for i in range(4)] + [corpus.text(20_000, seed=500 + i) for i in range(60)]
    data = b"".join(gzip.compress(p, 6) for p in plain)
    want = _decode(one, _lib.FMT_GZIP, data)
    assert want[0] == 0 and want[3] == b"".join(plain)
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_GZIP, data) == want
        assert eng.get_option("uploaded-bytes") <= len(data) + eng.device_count * (64 * 1024 + 64)


def test_gzip_damaged_member_in_second_range(engines):
    one, two, three = engines
    plain = _members(40, 96 * 1024, 200)
    blobs = [bytearray(gzip.compress(p, 6)) for p in plain]
    blobs[27][len(blobs[27]) // 2] ^= 0x55                       # inside the deflate data of a member of a later range
    data = b"".join(bytes(b) for b in blobs)
    want = _decode(one, _lib.FMT_GZIP, data)
    assert want[0] != 0 and want[3] == b"".join(plain[:27])      # first error wins, earlier members stay
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_GZIP, data) == want


def test_bzip2_blocks_spread(engines):
    one, two, three = engines
    plain = corpus.text(5 * (1 << 20) + 12345, seed=7)
    data = bz2.compress(plain, 1) + bz2.compress(plain[:500000], 2)     # ~33 + ~3 blocks, two streams
    assert len(data) > (1 << 20)
    want = _decode(one, _lib.FMT_BZIP2, data)
    assert want[0] == 0 and want[3] == plain + plain[:500000]
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_BZIP2, data) == want


def test_bzip2_damaged_block_spread(engines):
    one, two, three = engines
    plain = corpus.text(5 * (1 << 20), seed=8)
    data = bytearray(bz2.compress(plain, 1))
    data[len(data) * 3 // 4] ^= 0x10
    data = bytes(data)
    want = _decode(one, _lib.FMT_BZIP2, data)
    assert want[0] != 0
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_BZIP2, data) == want


def test_xz_blocks_spread(engines):
    one, two, three = engines
    import subprocess
    plain = corpus.x86ish(6 << 20, seed=9)
    data = subprocess.run(["xz", "-c", "-T1", "--block-size=256KiB", "--x86", "--lzma2=preset=1", "--check=crc64"], input=plain,
                          stdout=subprocess.PIPE, check=True).stdout
    assert len(data) > (1 << 20)
    want = _decode(one, _lib.FMT_XZ, data)
    assert want[0] == 0 and want[3] == plain and lzma.decompress(data) == plain
    for eng in (two, three):
        assert _decode(eng, _lib.FMT_XZ, data) == want


def test_small_inputs_stay_on_the_primary(engines):
    one, two, _ = engines
    data = gzip.compress(b"hello " * 100) + gzip.compress(b"world " * 100)
    assert _decode(two, _lib.FMT_GZIP, data) == _decode(one, _lib.FMT_GZIP, data)


def test_one_shot_calls_through_a_multi_context(engines):
    one, two, _ = engines
    plain = _members(30, 128 * 1024, 300)
    data = b"".join(gzip.compress(p, 6) for p in plain)
    out, res, info = two.decode_stream(_lib.FMT_GZIP, data, True)
    assert res.status == 0 and out == b"".join(plain) and info.nmembers == 30
