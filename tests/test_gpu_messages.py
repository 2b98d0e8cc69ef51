"""GPU: a failed decode says what the reference says -- its own message text with its arguments (the `die` sites of
deflate.lisp, gzip.lisp, zlib.lisp, bzip2.lisp, xz.lisp), and after everything the reference would have delivered
before the bad data (interface.lisp:95-108).  The arguments are worked out here independently of the engine: from the
parameters the hand-built vectors of tests/vectors_reject.py were made with and from CPython's checksums."""
import bz2
import hashlib
import lzma
import zlib

import pytest

import corpus
import vectors_reject as vr
from decompress_amd import api

pytestmark = pytest.mark.gpu

FORMAT_NAMES = {0: "deflate", 1: "zlib", 2: "gzip", 3: "bzip2", 5: "lzma", 7: "lzma2", 8: "xz"}
BY_NAME = {v["name"]: v for v in vr.VECTORS}


def message_of(name):
    v = BY_NAME[name]
    with pytest.raises(api.DecompressionError) as e:
        api.decompress(FORMAT_NAMES[v["fmt"]], v["data"], all_members_p=v["all"])
    assert e.value.status == v["status"]
    return str(e.value)


def test_messages_with_two_arguments():
    hello = b"hello, hello, hello!"
    assert message_of("deflate: stored block LEN / NLEN mismatch") == \
        "Checksum mismatch in uncompressed block (required FFFC, got 1234)."
    assert message_of("deflate: distance reaches before the start of the data") == \
        "Reference points back further (4) than the window allows (3)."
    assert message_of("zlib: CINFO 0 window (256) enforced (README.org:184)") == \
        "Reference points back further (300) than the window allows (256)."
    g = BY_NAME["gzip: header CRC wrong"]["data"]
    assert message_of("gzip: header CRC wrong") == \
        "Header checksum mismatch (required BEEF, got %04X)." % (zlib.crc32(g[:10]) & 0xFFFF)
    assert message_of("gzip: data CRC wrong") == "Invalid data checksum (wanted 12345678, got %08X)." % zlib.crc32(hello)
    assert message_of("gzip: ISIZE wrong") == "Invalid data length check (wanted 21, got 20)."
    assert message_of("gzip: CRC and ISIZE both wrong: the CRC is reported (gzip.lisp:105-113)") == \
        "Invalid data checksum (wanted 00000001, got %08X)." % zlib.crc32(hello)
    assert message_of("bzip2: stream magic") == "Incorrect bzip2 magic bytes: 42 59"
    bz_plain = BY_NAME["bzip2: valid stream"]["out"]
    good = BY_NAME["bzip2: valid stream"]["data"]
    block_crc = int.from_bytes(good[10:14], "big")                    # "BZh1", 48-bit magic, then the block's CRC
    assert bz2.decompress(good) == bz_plain
    assert message_of("bzip2: block CRC") == "Invalid block CRC (expected DEADBEEF, got %08X)." % block_crc
    assert message_of("bzip2: stream CRC") == "Incorrect stream CRC (expected 0BADF00D, got %08X)" % block_crc
    assert message_of("xz: block check wrong") == "Incorrect block checksum (expected 04030201, got %08X)." % zlib.crc32(hello)
    import re
    assert re.fullmatch(r"Invalid block header CRC \(expected 04030201, got [0-9A-F]{8}\)\.", message_of("xz: block header CRC"))
    assert re.fullmatch(r"Incorrect index CRC \(expected 00000000, got [0-9A-F]{8}\)\.", message_of("xz: index CRC"))
    assert re.fullmatch(r"Incorrect footer CRC \(expected 01020304, got [0-9A-F]{8}\)\.", message_of("xz: footer CRC"))


def test_messages_with_one_argument_and_none():
    assert message_of("deflate: fixed block, distance code 30") == "Distance code out of bounds (0-29): 30"
    assert message_of("deflate: fixed block, lit/len code 287") == "Invalid literal/length code: 287"
    assert message_of("deflate: single distance code, coded as 1") == \
        "Unique distance code is out of bounds (0-29) or is not encoded as a zero bit."
    assert message_of("deflate: lit/len tree underfull") == "Underfull Huffman tree."
    assert message_of("zlib: compression method 7") == "Unsupported compression method: 7"
    assert message_of("zlib: CINFO 8") == "Invalid compression info (must be 0 - 7): 8"
    assert message_of("gzip: compression method 7") == "Unrecognized compression method: 7"
    assert message_of("bzip2: version byte") == "Unrecognized bzip version: 67"
    assert message_of("bzip2: level ':'") == "Invalid bzip2 compression level: 3A"
    assert message_of("bzip2: seven trees") == "Illegal tree count: 7"
    assert message_of("xz: check type 2 (reserved)") == "Unsupported checksum type: 2"
    # every vector that fails: the message is the reference's control string with arguments, never the bare fallback
    seen = 0
    for v in vr.VECTORS:
        if v["route"] != "stream" or v["status"] in (0, 1, 44) or v["status"] >= 200 or v["fmt"] not in FORMAT_NAMES:
            continue
        if v["status"] not in api.MESSAGES:
            continue                                                    # 75, 107: no message in the reference
        text = message_of(v["name"])
        assert "~" not in text and text.startswith(api.MESSAGES[v["status"]][0].split("~")[0]), v["name"]
        seen += 1
    assert seen > 80


def test_sha256_block_check_message():
    """xz.lisp:452-454 prints both checks with twice the check size as the width: 64 digits for SHA-256, the digest
    read as a little-endian integer (sha256.lisp:159-164)."""
    plain = corpus.text(70000, seed=31)
    good = lzma.compress(plain, lzma.FORMAT_XZ, lzma.CHECK_SHA256)
    index_size = 4 * (int.from_bytes(good[-8:-4], "little") + 1)
    at = len(good) - 12 - index_size - 32
    bad = bytearray(good)
    bad[at + 5] ^= 0x80
    with pytest.raises(api.DecompressionError) as e:
        api.decompress_all("xz", bytes(bad))
    assert e.value.status == 155
    stored = int.from_bytes(bad[at:at + 32], "little")
    real = int.from_bytes(hashlib.sha256(plain).digest(), "little")
    assert str(e.value) == "Incorrect block checksum (expected %064X, got %064X)." % (stored, real)


def test_bzip2_damaged_late_block_delivers_every_block_before_it():
    """The reference hands out a block's data as soon as the block is decoded: a damaged block 4 of a second stream
    comes after the whole first stream and blocks 0-3 of the second.  Through the member-walking stream (the state
    machine of lisp/gpu.lisp) and through decompress-all's result buffer alike."""
    from test_gpu_bzip2 import BLOCK_MAGIC, find_magic
    p1, p2 = corpus.text(450_000, seed=61), corpus.text(750_000, seed=62)
    s1, s2 = bz2.compress(p1, 1), bz2.compress(p2, 1)
    starts = find_magic(s2, BLOCK_MAGIC)
    assert len(starts) >= 7
    bad2 = bytearray(s2)
    bad2[(starts[4] + starts[5]) // 16] ^= 0x10                       # inside block 4's Huffman data
    data = s1 + bytes(bad2)
    s = api.make_full_decompression_stream("bzip2", data)
    assert s.header == {"block_size": 100000}
    got = b""
    with pytest.raises(api.DecompressionError) as e:
        while True:
            chunk = s.read(123_457)
            if not chunk:
                break
            got += chunk
    want = p1 + p2
    assert len(got) > len(p1) + 390_000 and got == want[:len(got)]      # four blocks of about 100 k (RLE1 aside)
    assert len(got) < len(p1) + 510_000
    assert e.value.status in (73, 10, 11, 67, 68, 69, 70, 71, 72, 1, 63) and "~" not in str(e.value)
    # the state machine, step by step: member 0 whole, a reset, then the good octets of the failed member, not final
    st = api.GpuBatchState("bzip2", data)
    assert len(st.members) == 1 and st.members[0] == (len(s1), len(p1)) and st.out_len == len(got)
    first = b""
    while True:
        chunk, finalp = st.next_decompressed_chunk()
        first += chunk
        if finalp:
            break
    assert first == p1 and not st.source_at_eof()
    assert st.make_reset_state() is st
    rest = b""
    with pytest.raises(api.DecompressionError):
        while True:
            chunk, finalp = st.next_decompressed_chunk()
            assert not finalp
            rest += chunk
    assert first + rest == got
    # a first member that fails: its good blocks, then the error (no member was completed)
    s = api.make_decompression_stream("bzip2", bytes(bad2))
    got2 = b""
    with pytest.raises(api.DecompressionError):
        while True:
            chunk = s.read(1 << 20)
            if not chunk:
                break
            got2 += chunk
    assert got2 == got[len(p1):] and len(got2) > 390_000


def test_single_member_read_ignores_what_follows():
    """make-decompression-stream without all-members-p: the first member only, whatever follows it -- even
    garbage -- is never looked at (interface.lisp:98-102), though the engine decoded (and failed on) it already."""
    import gzip
    p = corpus.text(90_000, seed=5)
    s = api.make_decompression_stream("gzip", gzip.compress(p) + b"this is not a gzip member")
    assert s.read() == p and s.read() == b""
    s = api.make_full_decompression_stream("gzip", gzip.compress(p) + b"this is not a gzip member")
    assert s.read() == p                                 # (what was decoded before the bad data is delivered first)
    with pytest.raises(api.DecompressionError) as e:
        s.read()
    assert str(e.value) == "Incorrect gzip magic bytes."
    s = api.make_full_decompression_stream("zlib", zlib.compress(p) + b"x")
    assert s.read() == p
    with pytest.raises(api.DecompressionError, match="Trailing data detected."):
        s.read()


def test_release_after_close_is_harmless():
    """A decoded stream that outlives its context (a finalizer that runs after close-engine): semz_close takes its
    device memory back, a later release touches no context, a later read fails cleanly."""
    import gzip
    from decompress_amd import _lib
    eng = _lib.Engine(0)
    d = eng.stream_decode(_lib.FMT_GZIP, gzip.compress(corpus.text(50_000, seed=8)), True)
    assert d.res.status == 0 and d.read(0, 10) == corpus.text(50_000, seed=8)[:10]
    handle_engine = d.engine
    eng.close()
    import ctypes as C
    buf = C.create_string_buffer(16)
    assert handle_engine.lib.semz_stream_read(d.h, 0, buf, 10) != 0
    d.release()
