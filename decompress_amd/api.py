"""Host-side mirror of the reference's public interface (interface.lisp:8-208,
common.lisp:3-19) on top of the C-ABI library.

The reference is Common Lisp and no Lisp implementation exists in the build
image, so the host side above the C ABI is written here in Python with the
reference's names, argument meaning and error behaviour; INTEGRATION.md shows
the CFFI glue a maintainer of the Lisp library would add instead.

    decompress(format, input, start=0, end=None, output_size=None,
               all_members_p=False, allow_overreads_p=False, **format_options)
        -> (bytes, header_dict)                      interface.lisp:149-186
    decompress_all(format, input, ...)               interface.lisp:188-208
    make_decompression_stream / make_full_decompression_stream
        -> DecompressionStream (a readable binary file object with .header)
    list_supported_formats()                         common.lisp:291-295
    make_simple_zlib_dictionary(buffers)             zlib.lisp:80-100
    adler_32(data, start=0, end=None)                adler32.lisp:76-87

Formats are the reference's keywords as strings: "deflate", "zlib", "gzip",
"bzip2", "xz", "raw-lzma2" (needs window_size=, lzma2.lisp:206-218), "lzma2",
"lzma" (.lzma files) and "raw-lzma" (needs lc=, lp=, pb=, window_size=; takes
decompressed_size= and eof_mode= "maybe" | "always" | "never", lzma.lisp:809-825).  Everything decodes on
the GPU; there is no CPU fallback: without the built library or without a GPU
the calls raise EngineError.
"""
import io
import zlib as _zlib

from . import _lib

FORMATS = {"deflate": _lib.FMT_DEFLATE, "zlib": _lib.FMT_ZLIB, "gzip": _lib.FMT_GZIP, "bzip2": _lib.FMT_BZIP2,
           "xz": _lib.FMT_XZ, "raw-lzma2": _lib.FMT_RAW_LZMA2, "lzma2": _lib.FMT_LZMA2,
           "raw-lzma": _lib.FMT_RAW_LZMA, "lzma": _lib.FMT_LZMA}


class DecompressionError(Exception):
    """`decompression-error` (common.lisp:40-54): malformed or unsupported input."""

    def __init__(self, status, detail=0, message=None):
        self.status = status
        self.detail = detail
        super().__init__(message or _lib.strerror(status))


class EOF(DecompressionError):
    """`eof` (common.lisp:43-57): the input ended prematurely."""


class UnrecognizedZlibDictionary(DecompressionError):
    """`unrecognized-zlib-dictionary` (zlib.lisp:7-19); .checksum is the Adler-32 the stream asks for."""

    @property
    def checksum(self):
        return self.detail


def _raise(status, detail):
    if status == 1:
        raise EOF(status, detail)
    if status == 44:
        raise UnrecognizedZlibDictionary(status, detail)
    raise DecompressionError(status, detail)


def list_supported_formats():
    return list(FORMATS)


def adler_32(data, start=0, end=None):
    """Adler-32 of data[start:end] (adler32.lisp:76-87); a host helper used for dictionary ids."""
    return _zlib.adler32(bytes(data[start:end]))


def make_simple_zlib_dictionary(buffers):
    """Dictionary function over a list of byte strings, looked up by Adler-32 (zlib.lisp:80-100)."""
    table = {adler_32(b): bytes(b) for b in buffers}
    return lambda checksum: table.get(checksum)


_engine = None


def _get_engine():
    global _engine
    if _engine is None:
        _engine = _lib.Engine(0)
    return _engine


def _normalize_bounds(data, start, end):
    n = len(data)
    end = n if end is None else end
    if not (0 <= start <= end <= n):                     # normalize-bounds, common.lisp:136-147: a user error
        raise ValueError(f"bounds {start}..{end} outside a sequence of length {n}")
    return bytes(data[start:end])


def _header(fmt, data, res, info):
    h = info.first_header
    if fmt == "zlib":
        return {"window_size": h[0], "level": h[1], "dictionary": h[3] if h[2] else None}
    if fmt == "gzip":
        def field(off, ln):
            return data[off:off + ln].decode("latin-1") if off != 0xFFFFFFFF else None

        def extra_fields(off, ln):                        # parse-gzip-extra-fields, gzip.lisp:23-29: ((id . bytes) ...)
            if off == 0xFFFFFFFF:
                return []
            raw, out, i = data[off:off + ln], [], 0
            while i < len(raw):
                n = int.from_bytes(raw[i + 2:i + 4], "little")
                out.append((raw[i:i + 2].decode("latin-1"), raw[i + 4:i + 4 + n]))
                i += 4 + n
            return out
        return {"textp": bool(h[0] & 1), "extra_fields": extra_fields(h[2], h[3]), "filename": field(h[4], h[5]),
                "comment": field(h[6], h[7]), "modification_time": h[1], "extra_flags": (h[0] >> 8) & 0xFF,
                "operating_system": (h[0] >> 16) & 0xFF}
    if fmt == "bzip2":
        return {"block_size": 100000 * h[0]}
    if fmt == "xz":
        return {"checksum_type": h[0]}
    if fmt == "lzma2":
        return {"window_size": h[0]}                      # lzma2.lisp:226-227
    if fmt == "lzma":                                     # lzma.lisp:860-865
        return {"lc": h[0], "lp": h[1], "pb": h[2], "window_size": h[3],
                "decompressed_size": (h[6] << 32 | h[5]) if h[4] else None}
    return None


_EOF_MODES = {None: 0, "maybe": 0, "always": 1, "never": 2}


def _format_options(format, window_size, kw=None):
    """Format keywords that reach the decoder: :window-size of :raw-lzma2 (lzma2.lisp:206-213) and the
    keywords of :raw-lzma with their check-type ranges (lzma.lisp:811-825)."""
    kw = kw or {}
    if format in ("raw-lzma2", "raw-lzma"):
        if not isinstance(window_size, int) or window_size < 0:       # check-type window-size unsigned-byte
            raise TypeError(f"{format} needs window_size= (an unsigned integer)")
        _get_engine().set_option("window-size", min(window_size, 0xFFFFFFFF))
    if format == "raw-lzma":
        eng = _get_engine()
        for name, hi in (("lc", 8), ("lp", 4), ("pb", 4)):
            v = kw.get(name)
            if not isinstance(v, int) or not 0 <= v <= hi:
                raise TypeError(f"raw-lzma needs {name}= in 0..{hi}")
            eng.set_option(name, v)
        size = kw.get("decompressed_size")
        if size is not None and (not isinstance(size, int) or size < 0):
            raise TypeError("decompressed_size must be None or an unsigned integer")
        eng.set_option("decompressed-size", -1 if size is None else size)
        if kw.get("eof_mode") not in _EOF_MODES:
            raise TypeError("eof_mode must be 'always', 'never' or 'maybe'")
        eng.set_option("eof-mode", _EOF_MODES[kw.get("eof_mode")])


def _zlib_with_dictionary(data, dictionary, window_size):
    """zlib member with FDICT: the dictionary function supplies the preset history (zlib.lisp:37-56)."""
    eng = _get_engine()
    unit = _lib.Unit(in_off=0, in_limit=len(data), out_off=0, kind=_lib.UNIT_ZLIB, flags=_lib.UF_HAVE_DICT_FN)
    dict_id = int.from_bytes(data[2:6], "big") if len(data) >= 6 else 0
    preset = dictionary(dict_id) if callable(dictionary) else None
    blob = data
    if preset is not None:
        unit.prefix_off, unit.prefix_len = len(data), len(preset)
        blob = data + bytes(preset)
    cap = max(65536, 6 * len(data))
    while True:
        unit.out_cap = cap
        out, res = eng.decode_batch(blob, [unit], cap)
        if res[0].status != 200:
            break
        cap *= 4
    return out.raw[:res[0].out_len], res[0]


def decompress(format, input, start=0, end=None, output_size=None, all_members_p=False, allow_overreads_p=False,
               dictionary=None, window_size=None, **format_options):
    """Decompresses one member (or all, with all_members_p) of `input` (bytes-like or a binary file object).

    Returns (bytes, header) like the reference's two values; raises DecompressionError / EOF.
    `output_size` is only a pre-sizing hint in the reference (README.org:635-636) and is ignored here.
    """
    if format not in FORMATS:
        raise ValueError(f"unknown format {format!r}; supported: {list_supported_formats()}")
    if hasattr(input, "read"):
        data = input.read()                               # streams ignore start/end (example.lisp:32)
    else:
        data = _normalize_bounds(input, start, end)
    if format == "zlib" and len(data) >= 2 and (data[1] & 32) and dictionary is not None:
        payload, r = _zlib_with_dictionary(data, dictionary, window_size)
        if r.status != 0:
            _raise(r.status, r.detail)
        used = (r.in_end_bit + 7) // 8
        if all_members_p and used != len(data):
            _raise(170, 0)
        return payload, {"window_size": 1 << ((data[0] >> 4) + 8), "level": data[1] >> 6, "dictionary": r.aux}
    _format_options(format, window_size, format_options)
    payload, res, info = _get_engine().decode_stream(FORMATS[format], data, all_members_p)
    if res.status != 0:
        _raise(res.status, res.detail)
    if hasattr(input, "seek") and not all_members_p:
        # `decompress` leaves trailing data intact for stream inputs (example.lisp:41-52)
        try:
            input.seek(res.in_end_bit // 8 - len(data), io.SEEK_CUR)
        except (OSError, io.UnsupportedOperation):
            pass
    return payload, _header(format, data, res, info)


def decompress_all(format, input, **kw):
    """`decompress` with all_members_p set (interface.lisp:188-208)."""
    kw["all_members_p"] = True
    return decompress(format, input, **kw)


class DecompressionStream(io.RawIOBase):
    """`decompression-stream` (interface.lisp:8-44): a readable binary stream of the decoded data.

    The reference decodes lazily chunk by chunk.  This engine decodes all units on the GPU when the
    stream is made, keeps the result in device memory (semz_stream_decode) and hands it out member by
    member as it is read (semz_stream_members / semz_stream_read): exactly what the `gpu-batch-state`
    of lisp/gpu.lisp does behind `next-decompressed-chunk` / `make-reset-state`.  An error in the data
    is raised by the read that would reach the failing member, after the bytes of the members before
    it have been delivered (interface.lisp:105-108).
    """

    def __init__(self, format, input, all_members_p=False, **kw):
        super().__init__()
        if hasattr(input, "read"):
            data = input.read()
        else:
            data = _normalize_bounds(input, kw.pop("start", 0), kw.pop("end", None))
        self.format = format
        if format not in FORMATS:
            raise ValueError(f"unknown format {format!r}; supported: {list_supported_formats()}")
        _format_options(format, kw.get("window_size"), kw)
        self._d = _get_engine().stream_decode(FORMATS[format], data, all_members_p)
        res = self._d.res
        self._pending = None if res.status == 0 else (res.status, res.detail)
        self.header = _header(format, data, res, self._d.info)
        self._pos = 0
        self._len = self._d.out_len

    def readable(self):
        return True

    def read(self, size=-1):
        left = self._len - self._pos
        n = left if size is None or size < 0 else min(size, left)
        chunk = self._d.read(self._pos, n) if n else b""
        self._pos += n
        if self._pending and (size is None or size < 0 or n < size) and not chunk:
            st, self._pending = self._pending, None
            self._d.release()
            _raise(*st)
        return chunk

    def readinto(self, b):
        chunk = self.read(len(b))
        b[:len(chunk)] = chunk
        return len(chunk)

    def close(self):
        if getattr(self, "_d", None) is not None:
            self._d.release()
        super().close()


def make_decompression_stream(format, input, **kw):
    return DecompressionStream(format, input, **kw)


def make_full_decompression_stream(format, input, **kw):
    kw["all_members_p"] = True
    return DecompressionStream(format, input, **kw)
