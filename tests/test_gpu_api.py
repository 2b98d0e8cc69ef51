"""GPU: the host-side mirror of the reference's public interface (decompress_amd/api.py):
same names, same results as example.lisp, same condition types."""
import io

import pytest

import corpus
from decompress_amd import api

pytestmark = pytest.mark.gpu

Z = bytes([120, 156, 99, 84, 100, 7, 0, 0, 79, 0, 42])
GZ2 = bytes([31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 102, 111, 111, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0,
             31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 98, 97, 114, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0])
MULTI = bytes([99, 84, 100, 7, 0, 99, 84, 123, 7, 0, 1, 2, 3])


def test_example_lisp():
    # example.lisp:6-7
    assert api.decompress_all("zlib", Z) == (b"\x01\x21\x07", {"window_size": 32768, "level": 2, "dictionary": None})
    # example.lisp:12-19
    out, hdr = api.decompress_all("gzip", GZ2)
    assert out == b"\x01\x21\x07" * 2 and hdr["filename"] == "foo" and hdr["comment"] is None
    assert (hdr["textp"], hdr["modification_time"], hdr["extra_flags"], hdr["operating_system"]) == (False, 0, 0, 0)
    # example.lisp:24-26
    assert api.decompress_all("deflate", Z, start=2, end=7) == (b"\x01\x21\x07", None)
    # example.lisp:29-30
    with pytest.raises(api.DecompressionError, match="Trailing data"):
        api.decompress_all("zlib", Z + b"\1\2\3")
    # example.lisp:45-52: decompress leaves trailing data of a stream intact
    f = io.BytesIO(MULTI)
    assert api.decompress("deflate", f)[0] == b"\x01\x21\x07"
    assert api.decompress("deflate", f)[0] == b"\x01\x26\xee"
    assert f.read() == b"\1\2\3"
    # example.lisp:58-61
    s = api.make_decompression_stream("zlib", Z)
    assert s.read() == b"\x01\x21\x07" and s.header["level"] == 2
    # example.lisp:67-69
    with pytest.raises(api.DecompressionError):
        api.decompress("deflate", MULTI, start=5, all_members_p=True)
    # example.lisp:72-74
    out, hdr = api.decompress_all("zlib", bytes([120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42]),
                                  dictionary=api.make_simple_zlib_dictionary([bytes([1, 33, 7])]))
    assert out == b"\x01\x21\x07" and hdr == {"window_size": 32768, "level": 3, "dictionary": 5177386}
    # example.lisp:77-78 (the formats this engine decodes)
    assert set(api.list_supported_formats()) <= {"xz", "lzma2", "raw-lzma2", "lzma", "raw-lzma", "bzip2", "gzip", "zlib", "deflate"}


def test_conditions():
    import gzip
    g = gzip.compress(corpus.text(5000, seed=1))
    with pytest.raises(api.EOF):
        api.decompress_all("gzip", g[:-3])
    with pytest.raises(api.UnrecognizedZlibDictionary) as e:
        api.decompress_all("zlib", bytes([120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42]),
                           dictionary=api.make_simple_zlib_dictionary([b"other"]))
    assert e.value.checksum == 5177386
    with pytest.raises(api.DecompressionError, match="dictionary function"):
        api.decompress_all("zlib", bytes([120, 249, 0, 79, 0, 42, 3, 34, 0, 0, 79, 0, 42]))
    with pytest.raises(ValueError):
        api.decompress("brotli", b"")
    # a stream delivers the good members before raising at the bad one (interface.lisp:105-108)
    bad = bytearray(g + g); bad[-5] ^= 1
    s = api.make_full_decompression_stream("gzip", bytes(bad))
    assert s.read(5000) == corpus.text(5000, seed=1)
    with pytest.raises(api.DecompressionError):
        s.read()
    assert api.adler_32(bytes([1, 33, 7])) == 5177386


def test_lzma2_formats():
    """:raw-lzma2 takes :window-size (lzma2.lisp:206-213); :lzma2 reports it in the header (lzma2.lisp:222-227)."""
    import lzma
    d = b"lzma2 through the public interface " * 300
    raw = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 2, "dict_size": 1 << 16}])
    out, header = api.decompress("raw-lzma2", raw, window_size=1 << 16)
    assert out == d and header is None
    out, _ = api.decompress("raw-lzma2", raw, window_size=1 << 40)      # clipped to 2^32 - 1
    assert out == d
    with pytest.raises(TypeError):
        api.decompress("raw-lzma2", raw)
    out, header = api.decompress("lzma2", bytes([10]) + raw)
    assert out == d and header == {"window_size": 1 << 17}
    with pytest.raises(api.DecompressionError) as e:
        api.decompress("lzma2", bytes([41]) + raw)
    assert e.value.status == 101
    with pytest.raises(api.DecompressionError) as e:
        api.decompress_all("lzma2", bytes([10]) + raw + b"\0")
    assert e.value.status == 170
    s = api.make_decompression_stream("raw-lzma2", raw, window_size=1 << 16)
    assert s.read() == d


def test_lzma_formats():
    """:lzma reports (:lc :lp :pb :window-size :decompressed-size), lzma.lisp:860-865; :raw-lzma takes them as keywords."""
    import lzma
    d = b".lzma through the public interface " * 400
    a = lzma.compress(d, lzma.FORMAT_ALONE, filters=[{"id": lzma.FILTER_LZMA1, "preset": 1, "lc": 2, "lp": 1, "pb": 0,
                                                        "dict_size": 1 << 16}])
    out, header = api.decompress("lzma", a)
    assert out == d and header == {"lc": 2, "lp": 1, "pb": 0, "window_size": 1 << 16, "decompressed_size": None}
    out, header = api.decompress("raw-lzma", a[13:], lc=2, lp=1, pb=0, window_size=1 << 16)
    assert out == d and header is None
    out, _ = api.decompress("raw-lzma", a[13:], lc=2, lp=1, pb=0, window_size=1 << 16, decompressed_size=len(d),
                            eof_mode="always")
    assert out == d
    with pytest.raises(api.DecompressionError) as e:
        api.decompress("raw-lzma", a[13:], lc=2, lp=1, pb=0, window_size=1 << 16, eof_mode="never")
    assert e.value.status == 88
    with pytest.raises(TypeError):
        api.decompress("raw-lzma", a[13:], lc=9, lp=1, pb=0, window_size=1 << 16)
    with pytest.raises(api.EOF):
        api.decompress("lzma", a[:40])
    assert {"lzma", "raw-lzma", "lzma2", "raw-lzma2"} <= set(api.list_supported_formats())


def test_gzip_header_plist_with_extra_fields():
    """(:textp :extra-fields :filename :comment :modification-time :extra-flags :operating-system), gzip.lisp:81-87;
    :extra-fields is the alist of parse-gzip-extra-fields (gzip.lisp:23-29)."""
    import zlib
    d = corpus.text(300, seed=9)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(d) + co.flush()
    extra = b"BC\x02\x00\x12\x34" + b"zz\x00\x00" + b"Ap\x03\x00xyz"
    h = b"\x1f\x8b\x08\x1d" + (1234567).to_bytes(4, "little") + b"\x02\x03"
    h += len(extra).to_bytes(2, "little") + extra + b"foo\0" + b"a comment\0"
    member = h + body + zlib.crc32(d).to_bytes(4, "little") + len(d).to_bytes(4, "little")
    out, hdr = api.decompress_all("gzip", member + member)
    assert out == d + d
    assert hdr == {"textp": True, "extra_fields": [("BC", b"\x12\x34"), ("zz", b""), ("Ap", b"xyz")],
                   "filename": "foo", "comment": "a comment", "modification_time": 1234567,
                   "extra_flags": 2, "operating_system": 3}
    _, hdr = api.decompress("gzip", GZ2)
    assert hdr["extra_fields"] == [] and hdr["filename"] == "foo"
    # through an offset view as well (the spans in first_header are relative to the decoded input)
    out, hdr = api.decompress_all("gzip", b"junk!" + member, start=5)
    assert out == d and hdr["extra_fields"][2] == ("Ap", b"xyz")


def test_stream_hands_out_members_in_order():
    """The state machine lisp/gpu.lisp runs behind next-decompressed-chunk / make-reset-state: members come out
    one after the other from device memory; the error of member k is raised by the read that reaches it."""
    import gzip
    from decompress_amd import _lib
    parts = [corpus.text(20000 + 3000 * i, seed=70 + i) for i in range(6)]
    members = [gzip.compress(p, 6) for p in parts]
    bad = bytearray(b"".join(members))
    off = sum(len(m) for m in members[:4]) + len(members[4]) // 2
    bad[off] ^= 0x20                                                 # damage inside member 4
    eng = api._get_engine()
    d = eng.stream_decode(_lib.FMT_GZIP, bytes(bad), True)
    assert d.res.status != 0 and len(d.members) == 4
    pos_in = pos_out = 0
    for k, (in_end, out_end) in enumerate(d.members):
        assert in_end == pos_in + len(members[k]) and out_end == pos_out + len(parts[k])
        assert d.read(pos_out, out_end - pos_out) == parts[k]       # ranged reads straight from the device
        pos_in, pos_out = in_end, out_end
    assert d.out_len == pos_out
    with pytest.raises(_lib.EngineError):
        d.read(pos_out, 1)                                           # nothing beyond the last good member
    d.release()
    s = api.make_full_decompression_stream("gzip", bytes(bad))
    got = b""
    with pytest.raises(api.DecompressionError):
        while True:
            chunk = s.read(7777)
            if not chunk:
                break
            got += chunk
    assert got == b"".join(parts[:4])
    # two streams alive at once do not disturb each other (handles, not a per-context singleton)
    s1 = api.make_full_decompression_stream("gzip", b"".join(members[:3]))
    s2 = api.make_full_decompression_stream("gzip", b"".join(members[3:]))
    a = s1.read(100); b = s2.read(100)
    assert a + s1.read() == b"".join(parts[:3]) and b + s2.read() == b"".join(parts[3:])


def test_decode_stream_into_caller_buffer():
    import ctypes
    import gzip
    from decompress_amd import _lib
    parts = [corpus.mixed(50000, seed=80 + i) for i in range(5)]
    data = b"".join(gzip.compress(p, 6) for p in parts)
    want = b"".join(parts)
    eng = api._get_engine()
    buf = ctypes.create_string_buffer(len(want) + 100)
    rc, n, res, info = eng.decode_stream_into(_lib.FMT_GZIP, data, buf, True)
    assert (rc, n, res.status, info.nmembers) == (0, len(want), 0, 5) and buf.raw[:n] == want
    assert buf.raw[n:] == b"\0" * 100                                # nothing written beyond the stream
    small = ctypes.create_string_buffer(len(want) - 1)
    rc, n, res, _ = eng.decode_stream_into(_lib.FMT_GZIP, data, small, True)
    assert (rc, n, res.status) == (1, len(want), 0) and small.raw == b"\0" * (len(want) - 1)


def test_damaged_member_late_in_a_large_stream_delivers_everything_before_it():
    """A stream large enough for the sliced download (4 x 16 MiB and more): the members in front of a damaged one
    arrive byte-exact in the caller's buffer, the status is the damaged member's, nothing is written behind."""
    import ctypes
    import gzip
    from decompress_amd import _lib
    parts = [corpus.text(2 << 20, seed=900 + i) for i in range(40)]
    members = [gzip.compress(p, 1) for p in parts]
    bad = bytearray(b"".join(members))
    k = 29                                                            # in the third quarter
    bad[sum(len(m) for m in members[:k]) + len(members[k]) // 2] ^= 0x04
    want = b"".join(parts[:k])
    eng = api._get_engine()
    buf = ctypes.create_string_buffer(len(b"".join(parts)) + 64)
    rc, n, res, info = eng.decode_stream_into(_lib.FMT_GZIP, bytes(bad), buf, True)
    assert rc == 0 and res.status != 0 and info.nmembers == k and n == len(want)
    assert buf.raw[:n] == want and buf.raw[n:n + 64] == b"\0" * 64
    good, res2, info2 = eng.decode_stream(_lib.FMT_GZIP, b"".join(members), True)
    assert res2.status == 0 and info2.nmembers == 40 and good == b"".join(parts)
