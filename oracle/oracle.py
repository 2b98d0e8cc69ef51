"""ctypes wrapper of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by decompress_amd/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")


class OResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail", C.c_uint32), ("in_end_bit", C.c_uint64),
                ("out_len", C.c_uint64), ("check", C.c_uint8 * 32), ("check_len", C.c_uint32),
                ("pad", C.c_uint32)]


class ZlibHeader(C.Structure):
    _fields_ = [("window_size", C.c_uint32), ("level", C.c_uint32), ("has_dict", C.c_uint32),
                ("dict_id", C.c_uint32)]


class GzipHeader(C.Structure):
    _fields_ = [("textp", C.c_uint32), ("mtime", C.c_uint32), ("extra_flags", C.c_uint32), ("os", C.c_uint32),
                ("extra_off", C.c_int64), ("extra_len", C.c_int64), ("name_off", C.c_int64),
                ("name_len", C.c_int64), ("comment_off", C.c_int64), ("comment_len", C.c_int64),
                ("header_len", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.orc_crc32.restype = C.c_uint32
        _lib.orc_crc64.restype = C.c_uint64
        _lib.orc_adler32.restype = C.c_uint32
        _lib.orc_bzcrc.restype = C.c_uint32
        for f in ("orc_crc32", "orc_crc64", "orc_adler32", "orc_bzcrc"):
            getattr(_lib, f).argtypes = [C.c_char_p, C.c_size_t]
        _lib.orc_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        _lib.orc_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint, C.c_uint32, C.c_char_p,
                                     C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(OResult)]
        _lib.orc_zlib_member.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p,
                                         C.c_size_t, C.POINTER(ZlibHeader), C.POINTER(OResult)]
        _lib.orc_gzip_member.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                         C.POINTER(GzipHeader), C.POINTER(OResult)]
        _lib.orc_bzip2_block.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint, C.c_char_p, C.c_size_t,
                                         C.POINTER(OResult)]
        _lib.orc_bzip2_member.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(OResult)]
        _lib.orc_lzma2.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32, C.c_char_p,
                                   C.c_size_t, C.POINTER(OResult)]
        _lib.orc_lzma.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint32, C.c_int, C.c_uint64,
                                  C.c_int, C.c_char_p, C.c_size_t, C.POINTER(OResult)]
        _lib.orc_lzma_alone.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t,
                                        C.POINTER(C.c_uint32 * 5), C.POINTER(C.c_uint64), C.POINTER(OResult)]
        _lib.orc_bcj_x86.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        _lib.orc_bcj_simple.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint]
        _lib.orc_delta.argtypes = [C.c_char_p, C.c_size_t, C.c_uint]
        _lib.orc_xz_member.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(OResult)]
        _lib.orc_decompress.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32, C.c_char_p,
                                        C.c_size_t, C.POINTER(OResult)]
    return _lib


def crc32(b): return lib().orc_crc32(bytes(b), len(b))
def crc64(b): return lib().orc_crc64(bytes(b), len(b))
def adler32(b): return lib().orc_adler32(bytes(b), len(b))
def bzcrc(b): return lib().orc_bzcrc(bytes(b), len(b))


def sha256(b):
    d = C.create_string_buffer(32)
    lib().orc_sha256(bytes(b), len(b), d)
    return d.raw


def _run(fn, cap, *args):
    out = C.create_string_buffer(max(cap, 1))
    r = OResult()
    fn(*args, out, cap, C.byref(r))
    return r, out.raw[:r.out_len]


def inflate(data, start_bit=0, trailing_bits=0, window=32768, prefix=None, cap=1 << 20):
    p = bytes(prefix) if prefix is not None else None
    return _run(lambda *a: lib().orc_inflate(bytes(data), len(data), start_bit, trailing_bits, window, p,
                                             len(p) if p else 0, *a), cap)


def zlib_member(data, have_dict_fn=False, dictionary=None, cap=1 << 20):
    h = ZlibHeader()
    d = bytes(dictionary) if dictionary is not None else None
    out = C.create_string_buffer(max(cap, 1))
    r = OResult()
    lib().orc_zlib_member(bytes(data), len(data), int(have_dict_fn), d, len(d) if d else 0, out, cap,
                          C.byref(h), C.byref(r))
    return r, out.raw[:r.out_len], h


def gzip_member(data, cap=1 << 20):
    h = GzipHeader()
    out = C.create_string_buffer(max(cap, 1))
    r = OResult()
    lib().orc_gzip_member(bytes(data), len(data), out, cap, C.byref(h), C.byref(r))
    return r, out.raw[:r.out_len], h


def bzip2_block(data, start_bit, level, cap=1 << 22):
    return _run(lambda *a: lib().orc_bzip2_block(bytes(data), len(data), start_bit, level, *a), cap)


def lzma2(data, dict_size, limit=None, cap=1 << 22):
    return _run(lambda *a: lib().orc_lzma2(bytes(data), len(data), 0 if limit is None else 1,
                                           limit or 0, dict_size, *a), cap)


def lzma1(data, lc, lp, pb, dict_size, size=None, eof_mode=0, cap=1 << 22):
    """:raw-lzma (lzma.lisp:809-835). eof_mode 0 :maybe 1 :always 2 :never; size = :decompressed-size or None."""
    return _run(lambda *a: lib().orc_lzma(bytes(data), len(data), lc, lp, pb, dict_size, 0 if size is None else 1,
                                          size or 0, eof_mode, *a), cap)


def lzma_alone(data, honour_size=False, cap=1 << 22):
    """:lzma, the .lzma container. Returns (result, output, header words [lc, lp, pb, window, have_size], size)."""
    hdr = (C.c_uint32 * 5)()
    hsize = C.c_uint64()
    r, out = _run(lambda *a: lib().orc_lzma_alone(bytes(data), len(data), int(honour_size), *a[:2], C.byref(hdr), C.byref(hsize), a[2]), cap)
    return r, out, list(hdr), hsize.value


def bcj_x86(buf, start=0):
    b = C.create_string_buffer(bytes(buf), len(buf))
    lib().orc_bcj_x86(b, len(buf), start)
    return b.raw[:len(buf)]


def bcj_simple(buf, filter_id, start=0):
    """Inverse of the xz BCJ filter with id 0x05 (PowerPC) .. 0x0A (ARM64) over the whole buffer."""
    b = C.create_string_buffer(bytes(buf), len(buf))
    lib().orc_bcj_simple(b, len(buf), start, filter_id)
    return b.raw[:len(buf)]


def delta(buf, distance):
    b = C.create_string_buffer(bytes(buf), len(buf))
    lib().orc_delta(b, len(buf), distance)
    return b.raw[:len(buf)]


def decompress(fmt, data, all_members=True, opt=0, cap=1 << 22):
    """Mirror of decompress / decompress-all. fmt: 0 deflate 1 zlib 2 gzip 3 bzip2 5 lzma (opt = 1: honour the declared size) 6 raw-lzma2 (opt = window size) 7 lzma2 8 xz."""
    return _run(lambda *a: lib().orc_decompress(fmt, bytes(data), len(data), int(all_members), opt, *a), cap)
