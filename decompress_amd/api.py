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


# status -> the reference's own format-control string and where its arguments come from; the same table as
# *messages* in lisp/gpu.lisp (tests/test_lisp_glue.py keeps the two equal and every string byte-identical to a `die`
# call of the reference).  Sources: "detail" = semz_result.detail; "aux" = semz_result.aux of a failed stream (a
# second argument that is not a checksum); "check" = the checksum the decoder computed, as the integer the reference
# compares; "high-octet" / "low-octet" = the two octets packed into detail; "stored-check" = the check field of the xz
# block as the input stores it, "hex-digits" = twice its size.
MESSAGES = {
    10: ("Underfull Huffman tree.",),
    11: ("Overfull Huffman tree.",),
    20: ("Block uses reserved BTYPE.",),
    21: ("Checksum mismatch in uncompressed block (required ~4,'0x, got ~4,'0x).", "aux", "detail"),
    22: ("Code lengths expand beyond bounds.",),
    23: ("Tried to repeat non-existent last code length.",),
    24: ("Distance code out of bounds (0-29): ~d", "detail"),
    25: ("Unique distance code is out of bounds (0-29) or is not ~\n                        encoded as a zero bit.",),
    26: ("Length code in literal-only block: ~d", "detail"),
    27: ("Reference points back further (~d) than the window allows (~d).", "detail", "aux"),
    28: ("Invalid literal/length code: ~d", "detail"),
    40: ("Unsupported compression method: ~X", "detail"),
    41: ("Invalid compression info (must be 0 - 7): ~X", "detail"),
    42: ("Invalid flag checksum.",),
    43: ("Cannot use preset dictionaries without a dictionary function.",),
    45: ("Invalid data checksum (wanted ~8,'0x, got ~8,'0x).", "detail", "check"),
    50: ("Incorrect gzip magic bytes.",),
    51: ("Unrecognized compression method: ~x", "detail"),
    52: ("Reserved flag bits are non-zero.",),
    53: ("Inconsistent lengths for gzip extra fields.",),
    54: ("Header checksum mismatch (required ~4,'0x, got ~4,'0x).", "detail", "aux"),
    55: ("Invalid data checksum (wanted ~8,'0x, got ~8,'0x).", "detail", "check"),
    56: ("Invalid data length check (wanted ~d, got ~d).", "detail", "aux"),
    60: ("Incorrect bzip2 magic bytes: ~2,'0x ~2,'0x", "high-octet", "low-octet"),
    61: ("Unrecognized bzip version: ~2,'0x", "detail"),
    62: ("Invalid bzip2 compression level: ~2,'0x", "detail"),
    63: ("Unrecognized magic bytes on block boundary.",),
    64: ("Block declares no used symbols.",),
    65: ("Illegal tree count: ~d", "detail"),
    66: ("Block defines no selectors.",),
    67: ("Invalid selector: ~d", "detail"),
    68: ("Code length goes out of bounds.",),
    69: ("Ran out of selectors before end of block.",),
    70: ("A/B run cannot possibly stay in bounds.",),
    71: ("A/B run expands beyond block size.",),
    72: ("Too many codes/expansions.",),
    73: ("Invalid block CRC (expected ~8,'0x, got ~8,'0x).", "detail", "check"),
    74: ("Incorrect stream CRC (expected ~8,'0x, got ~8,'0x)", "detail", "check"),
    80: ("LZMA data doesn't begin with a zero octet.",),
    81: ("Code starts out of range.",),
    82: ("Code goes out of range.",),
    83: ("Match extends beyond dictionary size.",),
    84: ("Match extends beyond available data.",),
    85: ("Match extends beyond declared decompressed size.",),
    86: ("Decompressed data goes beyond declared size.",),
    87: ("Didn't find mandatory EOF marker in LZMA data.",),
    88: ("Illegal EOF marker in LZMA data.",),
    89: ("LZMA data continues beyond declared size.",),
    90: ("Neither range decoder nor marker denote an end.",),
    91: ("Range decoder finishes in invalid state.",),
    92: ("Decompressed data ends before declared size.",),
    93: ("Property `pb' is out of bounds (0-4): ~d", "detail"),
    100: ("Property lc+lp out of bounds (0-4): ~d", "detail"),
    101: ("Unrecognized LZMA2 dictionary size byte.",),
    102: ("Didn't get required dictionary reset.",),
    103: ("Didn't get required property reset.",),
    104: ("Embedded LZMA data goes beyond LZMA2 block size.",),
    105: ("Illegal LZMA2 control byte: ~2,'0x", "detail"),
    106: ("Embedded LZMA data is shorter than declared.",),
    120: ("Invalid XZ magic bytes.",),
    121: ("Reserved header byte isn't zero.",),
    122: ("Incorrect header checksum.",),
    123: ("Reserved header flag bits aren't zero.",),
    124: ("Unsupported checksum type: ~X", "detail"),
    125: ("Trailing garbage data after XZ stream.",),
    126: ("XZ padding doesn't come in multiples of 4 bytes.",),
    127: ("Overlong multibyte integer encoding.",),
    128: ("Overlong multibyte integer.",),
    129: ("Index padding contains non-zero bytes.",),
    130: ("Index disagrees with the observed number of blocks.",),
    131: ("Index disagrees with the observed block sizes.",),
    132: ("Incorrect index CRC (expected ~8,'0x, got ~8,'0x).", "detail", "check"),
    133: ("Incorrect footer CRC (expected ~8,'0x, got ~8,'0x).", "detail", "check"),
    134: ("Incorrect backwards size field.",),
    135: ("Inconsistent stream flags between header and footer.",),
    136: ("Invalid footer magic bytes.",),
    137: ("Reserved block header bits aren't zero.",),
    138: ("Block header is truncated.",),
    139: ("Block header padding contains non-zero bytes.",),
    140: ("Embedded compressed data goes beyond declared XZ block size.",),
    141: ("The Delta filter must not come last.",),
    142: ("Invalid property length for Delta filter: ~d", "detail"),
    143: ("BCJ filters must not come last.",),
    144: ("Invalid property length for BCJ filter: ~d", "detail"),
    145: ("Offset ~8,0x is not aligned to ~d", "detail", "aux"),
    146: ("The LZMA2 filter must always come last.",),
    147: ("Invalid property length for LZMA2 filter: ~d", "detail"),
    148: ("Reserved LZMA property bits aren't zero.",),
    149: ("Unrecognized filter type: ~2,'0x", "detail"),
    150: ("Invalid block header CRC (expected ~8,'0x, got ~8,'0x).", "detail", "check"),
    151: ("Total decompressed size exceeds 2^63 - 1 bytes.",),
    152: ("XZ block is smaller than its declared size.",),
    153: ("Decompressed size doesn't match declared decompressed size.",),
    154: ("Block padding doesn't consist of null bytes.",),
    155: ("Incorrect block checksum (expected ~v,'0x, got ~v,'0x).", "hex-digits", "stored-check", "hex-digits", "check"),
    156: ("XZ stream is longer than 2^63 - 1 bytes.",),
    170: ("Trailing data detected.",),
}


def lisp_format(control, args):
    """The part of Common Lisp's `format` that the reference's messages use: ~d, ~x / ~X with optional
    mincol and pad character (~8,'0x; `v` takes mincol from the arguments), ~a, and tilde-newline."""
    out, i, args = [], 0, list(args)
    while i < len(control):
        c = control[i]
        if c != "~":
            out.append(c)
            i += 1
            continue
        i += 1
        if control[i] == "\n":                            # tilde-newline: the newline and the blanks after it vanish
            i += 1
            while i < len(control) and control[i] in " \t":
                i += 1
            continue
        params = []
        while True:
            if control[i] == "'":
                params.append(control[i + 1])
                i += 2
            elif control[i] in "vV":
                params.append(args.pop(0))
                i += 1
            elif control[i].isdigit():
                j = i
                while control[j].isdigit():
                    j += 1
                params.append(int(control[i:j]))
                i = j
            elif control[i] == ",":
                if not params:
                    params.append(None)
                i += 1
                continue
            else:
                break
            if control[i] == ",":
                i += 1
                continue
        d = control[i].lower()
        i += 1
        v = args.pop(0)
        text = {"d": lambda: str(v), "x": lambda: format(v, "X"), "a": lambda: str(v)}[d]()
        mincol = params[0] if params and params[0] is not None else 0
        pad = str(params[1]) if len(params) > 1 and params[1] is not None else " "
        out.append(text.rjust(mincol, pad[0]))
    return "".join(out)


class Failure:
    """Where a decoded stream stopped: the status and the arguments of the reference's message for it
    (the `failure` structure of lisp/gpu.lisp)."""

    def __init__(self, status, detail=0, aux=0, check=0, check_size=0, stored_check=0):
        self.status, self.detail, self.aux = status, detail, aux
        self.check, self.check_size, self.stored_check = check, check_size, stored_check

    @classmethod
    def from_result(cls, res, data=b""):
        """From the semz_result of a stream call; `data` is the input (the stored check of an xz block is read from it)."""
        n = min(int(res.check_len), 32)
        check = int.from_bytes(bytes(res.check[:n]), "little")
        stored, end = 0, int(res.in_end_bit) // 8
        if res.status == 155 and n <= end <= len(data):    # the block's check field ends where the decode stopped (xz.lisp:444-450)
            stored = int.from_bytes(bytes(data[end - n:end]), "little")
        return cls(res.status, res.detail, res.aux, check, n, stored)

    def message(self):
        entry = MESSAGES.get(self.status)
        if entry is None:                                  # no message in the reference: the engine's text
            return _lib.strerror(self.status)
        src = {"detail": self.detail, "aux": self.aux, "check": self.check, "high-octet": (self.detail >> 8) & 0xFF,
               "low-octet": self.detail & 0xFF, "hex-digits": 2 * self.check_size, "stored-check": self.stored_check}
        return lisp_format(entry[0], [src[a] for a in entry[1:]])


class DecompressionError(Exception):
    """`decompression-error` (common.lisp:40-54): malformed or unsupported input, with the reference's message."""

    def __init__(self, status, detail=0, message=None):
        self.status = status
        self.detail = detail
        super().__init__(message or Failure(status, detail).message())


class EOF(DecompressionError):
    """`eof` (common.lisp:43-57): the input ended prematurely."""


class UnrecognizedZlibDictionary(DecompressionError):
    """`unrecognized-zlib-dictionary` (zlib.lisp:7-19); .checksum is the Adler-32 the stream asks for."""

    @property
    def checksum(self):
        return self.detail


def _signal(failure):
    """signal-failure of lisp/gpu.lisp: the reference's condition for the status, with its message."""
    st = failure.status
    if st == 1:
        raise EOF(st, failure.detail, "Premature end of input.")               # common.lisp:56-57
    if st == 44:
        raise UnrecognizedZlibDictionary(st, failure.detail, _lib.strerror(st))  # zlib.lisp:47-48
    if st >= 200:
        raise _lib.EngineError(_lib.strerror(st))          # engine-level, no counterpart in the reference
    raise DecompressionError(st, failure.detail, failure.message())


def _raise(status, detail, res=None, data=b""):
    _signal(Failure.from_result(res, data) if res is not None else Failure(status, detail))


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
    return out.raw[:res[0].out_len], res[0], len(preset) if preset is not None else 0


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
        payload, r, history = _zlib_with_dictionary(data, dictionary, window_size)
        if r.status != 0:
            f = Failure.from_result(r, data)
            # a unit's result carries the first argument only; the second ones of these messages (stream.cpp: fail_from_unit)
            if r.status == 27:                            # the reach the window allowed (deflate.lisp:330): history included
                f.aux = min(1 << ((data[0] >> 4) + 8), r.out_len + history)
            elif r.status == 21:                          # the value NLEN should have had (deflate.lisp:384)
                end = (r.in_end_bit + 7) // 8
                f.aux = int.from_bytes(data[end - 4:end - 2], "little") ^ 0xFFFF
            _signal(f)
        used = (r.in_end_bit + 7) // 8
        if all_members_p and used != len(data):
            _raise(170, 0)
        return payload, {"window_size": 1 << ((data[0] >> 4) + 8), "level": data[1] >> 6, "dictionary": r.aux}
    _format_options(format, window_size, format_options)
    payload, res, info = _get_engine().decode_stream(FORMATS[format], data, all_members_p)
    if res.status != 0:
        _raise(res.status, res.detail, res, data)
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


_MULTI_MEMBER_FORMATS = ("gzip", "bzip2", "xz")              # gzip.lisp:92-93, bzip2.lisp:453-454, xz.lisp:118-134
# statuses the reference signals while it reads a member's header, i.e. inside byte-source->decompression-state
_HEADER_STATUSES = {40, 41, 42, 43, 44, 50, 51, 52, 53, 54, 60, 61, 62, 93, 101, 120, 121, 122, 123, 124}
CHUNK_SIZE = 4 << 20                                          # *chunk-size* of lisp/gpu.lisp (common.lisp:255-257)


class GpuBatchState:
    """The `gpu-batch-state` of lisp/gpu.lisp, statement for statement: all members of one input, decoded by the
    engine in one call and waiting in device memory (semz_stream_decode), handed out through the reference's format
    protocol (common.lisp:262-289) -- `next_decompressed_chunk` a member at a time in bounded chunks with `finalp`
    at the end of each member, `make_reset_state` on to the next member.  The octets the member that failed had
    delivered before its bad data come out (not final) before the failure is signalled (interface.lisp:95-108)."""

    def __init__(self, format, data):
        self.format = format
        self.data = data
        # multi-member formats are decoded to the end: which members the caller wants is decided later, by
        # whether it calls make_reset_state (all-members-p never reaches the format protocol, interface.lisp:85)
        self.decoded = _get_engine().stream_decode(FORMATS[format], data, format in _MULTI_MEMBER_FORMATS)
        res = self.decoded.res
        self.members = self.decoded.members               # [(in_end, out_end)] of the members the decode completed
        self.consumed = res.in_end_bit // 8
        self.out_len = res.out_len
        self.index = 0                                    # the member being handed out
        self.offset = 0                                   # octets of output handed out so far
        self.source_pos = 0                               # `buffer-stream-start` of the byte source (advance-source)
        self.pending = Failure.from_result(res, data) if res.status != 0 else None
        self.header = _header(format, data, res, self.decoded.info)
        if self.pending and not self.members and res.status in _HEADER_STATUSES:
            self.release()                                # a header the reference rejects is rejected where it rejects it
            _signal(self.pending)

    def release(self):
        if self.decoded is not None:
            self.decoded.release()
            self.decoded = None

    def _fetch(self, n):
        return self.decoded.read(self.offset, n) if n else b""

    def next_decompressed_chunk(self):
        """-> (chunk, finalp)"""
        k = self.index
        if k < len(self.members):
            in_end, out_end = self.members[k]
            n = min(CHUNK_SIZE, out_end - self.offset)
            chunk = self._fetch(n)
            self.offset += n
            finalp = self.offset == out_end
            if finalp:
                self.source_pos = min(len(self.data), in_end)
                if k + 1 == len(self.members) and self.pending is None:
                    self.release()
            return chunk, finalp
        if self.pending and self.offset < self.out_len:   # the good octets of the member that failed
            n = min(CHUNK_SIZE, self.out_len - self.offset)
            chunk = self._fetch(n)
            self.offset += n
            return chunk, False
        if self.pending:
            self.release()
            _signal(self.pending)
        return b"", True

    def make_reset_state(self):
        if self.format not in _MULTI_MEMBER_FORMATS:      # the caller says "Trailing data detected." (interface.lisp:125-130)
            return None
        self.index += 1
        if self.index >= len(self.members) and self.pending is None:
            # nothing but what the engine already accepted is left (xz stream padding, xz.lisp:118-134)
            self.source_pos = min(len(self.data), self.consumed)
        return self

    def source_at_eof(self):                              # buffer-stream-check-eof, interface.lisp:110-112
        return self.source_pos >= len(self.data)


class DecompressionStream(io.RawIOBase):
    """`decompression-stream` (interface.lisp:8-44): a readable binary stream of the decoded data, driven by the
    `refill` state machine of interface.lisp:91-133 over a format state.  The state is a `GpuBatchState`: an
    error in the data is raised by the read that reaches the bad data, after everything decoded before it has
    been delivered (interface.lisp:105-108)."""

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
        self.all_members_p = all_members_p
        self._state = GpuBatchState(format, data)
        self.header = self._state.header
        self._control = "has-member"
        self._buf = b""
        self._error = None

    def readable(self):
        return True

    def _refill(self):
        """One step of `refill`: a chunk (possibly empty), or None at the end of the stream."""
        if self._control == "eof":
            return None
        if self._control == "has-member":
            chunk, finalp = self._state.next_decompressed_chunk()
            if finalp:
                self._control = "needs-member" if self.all_members_p else "eof"
            return chunk
        # needs-member
        if self._state.source_at_eof():
            self._control = "eof"
            return None
        new_state = self._state.make_reset_state()
        if new_state is None:
            _signal(Failure(170))                         # "Trailing data detected."
        self._state = new_state
        self._control = "has-member"
        return b""

    def read(self, size=-1):
        """Up to `size` octets (all that is left when negative).  Bad data raises from the read that reaches it; when
        that read has octets to deliver that were decoded before the bad data, it returns them and the next read raises."""
        want = None if size is None or size < 0 else size
        parts, got = [], 0
        if self._buf:
            take = self._buf if want is None else self._buf[:want]
            self._buf = self._buf[len(take):]
            parts.append(take)
            got += len(take)
        while (want is None or got < want) and self._error is None:
            try:
                chunk = self._refill()
            except DecompressionError as e:
                self._error = e
                break
            if chunk is None:
                break
            if want is not None and got + len(chunk) > want:
                cut = want - got
                chunk, self._buf = chunk[:cut], chunk[cut:]
            parts.append(chunk)
            got += len(chunk)
        if self._error is not None and not got and want != 0:
            e, self._error = self._error, None
            raise e
        return b"".join(parts)

    def readinto(self, b):
        chunk = self.read(len(b))
        b[:len(chunk)] = chunk
        return len(chunk)

    def close(self):
        st = getattr(self, "_state", None)
        if st is not None:
            st.release()
        super().close()


def make_decompression_stream(format, input, **kw):
    return DecompressionStream(format, input, **kw)


def make_full_decompression_stream(format, input, **kw):
    kw["all_members_p"] = True
    return DecompressionStream(format, input, **kw)
