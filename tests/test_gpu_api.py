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
