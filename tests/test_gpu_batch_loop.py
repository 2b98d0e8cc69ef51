"""GPU: the batch loop of the stream calls (/root/reference/interface.lisp:91-133,
common.lisp:255-257).  With a small "batch-budget" a predicted batch does not fit the device in one piece and is
decoded window by window into host memory, two windows in flight.  Whatever the budget, a call must return what the
one-piece call returns: bytes, status, detail, consumed input, member table -- and when a member in a late window is
damaged, everything before it byte for byte."""
import bz2
import ctypes as C
import gzip
import lzma
import struct
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


def _into(eng, fmt, data, cap):
    buf = C.create_string_buffer(max(cap, 1))
    rc, n, res, info = eng.decode_stream_into(fmt, data, buf)
    return rc, res.status, res.detail, res.in_end_bit, (buf.raw[:n] if rc == 0 else None), n


@pytest.fixture(scope="module")
def engines():
    whole = _lib.Engine(0)
    tight = _lib.Engine(0)
    tight.set_option("batch-budget", 3 << 20)                    # windows of about 1.5 MiB of input + output
    assert tight.get_option("batch-budget") == 3 << 20 and whole.get_option("batch-budget") == 0
    yield whole, tight
    whole.close()
    tight.close()


def _members(n, size, seed):
    return [corpus.text(size + 977 * (i % 5), seed=seed + i) for i in range(n)]


def _check_all_ways(whole, tight, fmt, data, plain_want=None, min_windows=3):
    want = _decode(whole, fmt, data)
    assert whole.get_option("windows") == 0
    if plain_want is not None:
        assert want[0] == 0 and want[3] == plain_want
    got = _decode(tight, fmt, data)                               # windows in host memory of the library
    assert tight.get_option("windows") >= min_windows
    assert got == want
    n = len(want[3])
    rc, st, det, end, out, m = _into(tight, fmt, data, n + 100)   # windows straight into the caller's buffer
    assert tight.get_option("windows") >= min_windows
    assert (rc, st, det, end, out, m) == (0, want[0], want[1], want[2], want[3], n)
    if n:
        rc, *_rest, m = _into(tight, fmt, data, n - 1)            # a buffer too small: the size comes back, nothing else
        assert rc == 1 and m == n
    payload, res, _info = tight.decode_stream(fmt, data)
    assert (res.status, res.detail, res.in_end_bit, payload) == want[:4]
    return want


def test_gzip_members_in_windows(engines):
    whole, tight = engines
    plain = _members(40, 200 * 1024, 100)
    data = b"".join(gzip.compress(p, 6) for p in plain)
    want = _check_all_ways(whole, tight, _lib.FMT_GZIP, data, b"".join(plain), min_windows=5)
    assert len(want[4]) == 40
    rc, st, *_ = _into(tight, _lib.FMT_GZIP, data, sum(map(len, plain)))
    assert (rc, st) == (0, 0) and tight.get_option("moved") == 0        # every window went where the stream has it


def test_gzip_damaged_member_in_a_late_window(engines):
    whole, tight = engines
    plain = _members(40, 200 * 1024, 200)
    blobs = [bytearray(gzip.compress(p, 6)) for p in plain]
    blobs[27][len(blobs[27]) // 2] ^= 0x55                       # (about the sixth window of eight)
    data = b"".join(bytes(b) for b in blobs)
    want = _check_all_ways(whole, tight, _lib.FMT_GZIP, data, None, min_windows=5)
    assert want[0] != 0 and want[3] == b"".join(plain[:27]) and len(want[4]) == 27     # first error wins, earlier members stay


def test_gzip_trailer_damage_and_garbage_behind(engines):
    whole, tight = engines
    plain = _members(24, 160 * 1024, 300)
    blobs = [bytearray(gzip.compress(p, 6)) for p in plain]
    blobs[20][-6] ^= 1                                            # CRC-32 field of a late member
    data = b"".join(bytes(b) for b in blobs)
    want = _check_all_ways(whole, tight, _lib.FMT_GZIP, data)
    assert want[0] != 0 and want[3] == b"".join(plain[:20])
    data = b"".join(gzip.compress(p, 6) for p in plain) + b"\0" * 5000
    want = _check_all_ways(whole, tight, _lib.FMT_GZIP, data)
    assert want[0] != 0 and want[3] == b"".join(plain)


def test_gzip_look_alike_header_moves_the_layout(engines):
    """A stored block that holds the ten bytes of a member header: the candidate scan takes it for a member start,
    the member around it gets a span from the wrong ISIZE and is decoded again by the walk -- the windows that are
    already in the caller's buffer are then not where the stream has them."""
    whole, tight = engines
    plain = _members(30, 150 * 1024, 400)
    fake = bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 3]) + bytes(40)
    # (two of them: the first one's span ends in front of the second, where "its ISIZE" says 1000 bytes, so every
    # later member is predicted 1000 bytes further on than the stream has it)
    odd = plain[9][:70000] + fake + struct.pack("<I", 1000) + fake + plain[9][70000:]
    co = zlib.compressobj(0, zlib.DEFLATED, -15)                  # stored blocks: the bytes appear as they are
    raw = co.compress(odd) + co.flush()
    member = bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 3]) + raw + struct.pack("<II", zlib.crc32(odd), len(odd))
    blobs = [gzip.compress(p, 6) for p in plain]
    blobs[9] = member
    plain[9] = odd
    data = b"".join(blobs)
    _check_all_ways(whole, tight, _lib.FMT_GZIP, data, b"".join(plain))
    rc, st, _d, _e, out, _n = _into(tight, _lib.FMT_GZIP, data, sum(map(len, plain)) + 4096)   # (room for the prediction)
    assert (rc, st) == (0, 0) and out == b"".join(plain) and tight.get_option("moved") >= 1


def test_bgzf_members_in_windows(engines):
    whole, tight = engines
    plain = corpus.text(6 * (1 << 20) + 321, seed=5)
    blobs = []
    for i in range(0, len(plain), 60000):                         # BGZF: an extra field "BC" with the member's size
        piece = plain[i:i + 60000]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(raw) + 8 - 1
        blobs.append(bytes([0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize) + raw +
                     struct.pack("<II", zlib.crc32(piece), len(piece)))
    data = b"".join(blobs)
    _check_all_ways(whole, tight, _lib.FMT_GZIP, data, plain)


def test_bzip2_blocks_in_windows():
    whole = _lib.Engine(0)
    tight = _lib.Engine(0)
    tight.set_option("batch-budget", 96 << 20)                    # (a block takes ten megabytes of decoder scratch)
    try:
        plain = corpus.text(5 * (1 << 20) + 12345, seed=7)
        data = bz2.compress(plain, 1) + bz2.compress(plain[:500000], 2)     # ~53 + ~3 blocks, two streams
        _check_all_ways(whole, tight, _lib.FMT_BZIP2, data, plain + plain[:500000], min_windows=5)
        # long runs: RLE1 makes far more of a block than the estimate a window's room starts from
        runs = b"".join(bytes([i % 251]) * 255 for i in range(40000))
        data = bz2.compress(runs, 9)
        tiny = _lib.Engine(0)
        tiny.set_option("batch-budget", 8 << 20)                  # (less than one block's scratch: a window per block)
        try:
            _check_all_ways(whole, tiny, _lib.FMT_BZIP2, data, runs, min_windows=1)
        finally:
            tiny.close()
        # a damaged block late in the stream
        blob = bytearray(bz2.compress(plain, 1))
        blob[len(blob) * 3 // 4] ^= 0x10
        want = _check_all_ways(whole, tight, _lib.FMT_BZIP2, bytes(blob), None, min_windows=5)
        assert want[0] != 0 and 0 < len(want[3]) < len(plain) and plain.startswith(want[3])
    finally:
        whole.close()
        tight.close()


def test_xz_blocks_in_windows(engines):
    whole, tight = engines
    plain = corpus.text(6 * (1 << 20) + 77, seed=11)
    # (liblzma writes one block per stream: several streams give the units)
    streams = [lzma.compress(plain[i:i + 300000], lzma.FORMAT_XZ, check=lzma.CHECK_CRC32, preset=1)
               for i in range(0, len(plain), 300000)]
    data = b"".join(streams)
    _check_all_ways(whole, tight, _lib.FMT_XZ, data, plain, min_windows=5)
    bad = bytearray(data)
    at = sum(len(s) for s in streams[:15]) + len(streams[15]) // 2
    bad[at] ^= 0x40
    want = _check_all_ways(whole, tight, _lib.FMT_XZ, bytes(bad), None, min_windows=5)
    assert want[0] != 0 and want[3] == plain[:15 * 300000]


def test_an_input_larger_than_the_budget_is_never_resident(engines):
    """64 MiB of output from an input of 20 MiB under a 12 MiB budget: the input goes up in windows only."""
    whole, _ = engines
    eng = _lib.Engine(0)
    eng.set_option("batch-budget", 12 << 20)
    try:
        plain = _members(64, 1 << 20, 900)
        data = b"".join(gzip.compress(p, 1) for p in plain)
        assert len(data) > (12 << 20)
        rc, st, det, end, out, n = _into(eng, _lib.FMT_GZIP, data, sum(map(len, plain)))
        assert (rc, st, end) == (0, 0, 8 * len(data)) and out == b"".join(plain)
        assert eng.get_option("windows") >= 8
    finally:
        eng.close()


def test_sizes_that_lie_under_a_tight_budget():
    """The forged xz indexes and BGZF ISIZE fields of test_gpu_untrusted_sizes, with a budget that sends every
    prediction through the batch loop: still rc 0, the oracle's status and detail, the bytes before the error."""
    import test_gpu_untrusted_sizes as lies
    eng = _lib.Engine(0)
    eng.set_option("batch-budget", 1 << 18)
    try:
        lies.test_xz_index_that_lies(eng)
        lies.test_bgzf_isize_that_lies(eng)
        lies.test_block_header_sizes_that_lie(eng)
    finally:
        eng.close()


def test_windows_on_the_devices_of_a_multi_context(engines):
    """semz_open_multi + a tight budget: the helper devices are lanes of the batch loop beside the two of the primary
    device (the box has one GPU, so it is named three times)."""
    whole, _ = engines
    eng = _lib.Engine([0, 0, 0])
    eng.set_option("batch-budget", 3 << 20)
    try:
        plain = _members(48, 200 * 1024, 700)
        blobs = [bytearray(gzip.compress(p, 6)) for p in plain]
        data = b"".join(bytes(b) for b in blobs)
        want = _decode(whole, _lib.FMT_GZIP, data)
        assert want[0] == 0 and want[3] == b"".join(plain)
        assert _decode(eng, _lib.FMT_GZIP, data) == want and eng.get_option("windows") >= 6
        blobs[40][len(blobs[40]) // 2] ^= 0x55
        data = b"".join(bytes(b) for b in blobs)
        want = _decode(whole, _lib.FMT_GZIP, data)
        assert want[0] != 0 and want[3] == b"".join(plain[:40])
        assert _decode(eng, _lib.FMT_GZIP, data) == want
        rc, st, det, end, out, n = _into(eng, _lib.FMT_GZIP, data, len(want[3]) + (1 << 20))
        assert (rc, st, det, end, out) == (0, want[0], want[1], want[2], want[3])
        big = corpus.text(5 * (1 << 20), seed=70)
        data = bz2.compress(big, 1)
        eng.set_option("batch-budget", 96 << 20)
        assert _decode(eng, _lib.FMT_BZIP2, data) == _decode(whole, _lib.FMT_BZIP2, data)
        assert eng.get_option("windows") >= 4
    finally:
        eng.close()
