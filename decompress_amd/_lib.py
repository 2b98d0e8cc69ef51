"""ctypes binding of the C-ABI library (include/semz_gpu.h).

The library is the product: there is no Python or CPU decode path behind it.
Importing this module without the built ``libsemz_gpu.so`` raises, and opening
a context without a GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEMZ_LIB selects another build of the same library (e.g. the diagnostic libsemz_gpu_stamps.so)
LIB_PATH = os.environ.get("SEMZ_LIB") or os.path.join(_HERE, "libsemz_gpu.so")


class Unit(C.Structure):
    _fields_ = [
        ("in_off", C.c_uint64), ("in_limit", C.c_uint64),
        ("out_off", C.c_uint64), ("out_cap", C.c_uint64),
        ("kind", C.c_uint32), ("flags", C.c_uint32),
        ("in_bit_off", C.c_uint32), ("param0", C.c_uint32),
        ("param1", C.c_uint32), ("param2", C.c_uint32),
        ("param3", C.c_uint32), ("filters", C.c_uint32),
        ("prefix_off", C.c_uint64), ("prefix_len", C.c_uint64),
        ("size_in", C.c_uint64), ("size_out", C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("detail", C.c_uint32),
        ("in_end_bit", C.c_uint64), ("out_len", C.c_uint64), ("out_off", C.c_uint64),
        ("check", C.c_uint8 * 32), ("check_len", C.c_uint32), ("aux", C.c_uint32),
    ]


class Member(C.Structure):
    _fields_ = [("in_end", C.c_uint64), ("out_end", C.c_uint64)]


class StreamInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("detail", C.c_uint32), ("total_out", C.c_uint64),
        ("first_header", C.c_uint32 * 8), ("nmembers", C.c_uint32), ("reserved", C.c_uint32),
    ]


# unit kinds / formats / flags (semz_gpu.h)
UNIT_DEFLATE, UNIT_ZLIB, UNIT_GZIP, UNIT_BZIP2_BLOCK, UNIT_XZ_BLOCK, UNIT_LZMA2, UNIT_LZMA, UNIT_LZMA_WIDE = range(8)
FMT_DEFLATE, FMT_ZLIB, FMT_GZIP, FMT_BZIP2, FMT_RAW_LZMA, FMT_LZMA, FMT_RAW_LZMA2, FMT_LZMA2, FMT_XZ = range(9)
UF_HAVE_DICT_FN, UF_HAVE_IN_SIZE, UF_HAVE_OUT_SIZE = 1, 4, 8
FILTER_DELTA, FILTER_X86 = 1, 2
FILTER_POWERPC, FILTER_IA64, FILTER_ARM, FILTER_ARMTHUMB, FILTER_SPARC, FILTER_ARM64 = 3, 4, 5, 6, 7, 8
CHECK_NONE, CHECK_CRC32, CHECK_CRC64, CHECK_SHA256 = 0, 1, 4, 10

OUT_AUTO = 0xFFFFFFFFFFFFFFFF

_lib = None

EXPORTS = [
    "semz_open", "semz_open_multi", "semz_device_count", "semz_close", "semz_last_error", "semz_strerror", "semz_decode_batch",
    "semz_decode_batch_device", "semz_decode_batch_device_noalloc", "semz_set_option", "semz_get_option", "semz_sync", "semz_stream",
    "semz_dev_alloc", "semz_dev_free", "semz_dev_upload", "semz_dev_download",
    "semz_event_create", "semz_event_record", "semz_event_elapsed_ms", "semz_event_destroy",
    "semz_split", "semz_free", "semz_decode_stream",
    "semz_stream_decode", "semz_stream_members", "semz_stream_read", "semz_stream_release", "semz_decode_stream_into",
    "semz_profile_count", "semz_profile_get", "semz_profile_reset",
]


def load():
    """Load libsemz_gpu.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    lib.semz_open.argtypes = [C.c_int, C.POINTER(vp)]
    lib.semz_open_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.semz_device_count.argtypes = [vp]
    lib.semz_close.argtypes = [vp]
    lib.semz_close.restype = None
    lib.semz_last_error.argtypes = [vp]
    lib.semz_last_error.restype = C.c_char_p
    lib.semz_strerror.argtypes = [C.c_int]
    lib.semz_strerror.restype = C.c_char_p
    lib.semz_decode_batch.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(Unit), C.c_size_t,
                                      vp, C.c_size_t, C.POINTER(Result)]
    lib.semz_decode_batch_device.argtypes = [vp, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp]
    lib.semz_decode_batch_device_noalloc.argtypes = lib.semz_decode_batch_device.argtypes
    lib.semz_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    lib.semz_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
    lib.semz_sync.argtypes = [vp]
    lib.semz_stream.argtypes = [vp]
    lib.semz_stream.restype = vp
    lib.semz_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.semz_dev_free.argtypes = [vp, vp]
    lib.semz_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    lib.semz_dev_download.argtypes = [vp, vp, vp, C.c_size_t]
    lib.semz_event_create.argtypes = [vp, C.POINTER(vp)]
    lib.semz_event_record.argtypes = [vp, vp]
    lib.semz_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    lib.semz_event_destroy.argtypes = [vp, vp]
    lib.semz_split.argtypes = [C.c_int, C.c_char_p, C.c_size_t, u64p, u64p, C.c_size_t,
                               C.POINTER(C.POINTER(Unit)), C.POINTER(C.c_size_t), C.POINTER(StreamInfo)]
    lib.semz_free.argtypes = [vp]
    lib.semz_free.restype = None
    lib.semz_decode_stream.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, C.c_int,
                                       C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(Result),
                                       C.POINTER(StreamInfo)]
    lib.semz_stream_decode.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp),
                                       C.POINTER(Result), C.POINTER(StreamInfo)]
    lib.semz_stream_members.argtypes = [vp, C.POINTER(C.POINTER(Member)), C.POINTER(C.c_size_t)]
    lib.semz_stream_read.argtypes = [vp, C.c_uint64, vp, C.c_size_t]
    lib.semz_stream_release.argtypes = [vp]
    lib.semz_stream_release.restype = None
    lib.semz_decode_stream_into.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, C.c_int, vp, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(Result), C.POINTER(StreamInfo)]
    lib.semz_profile_count.argtypes = [vp]
    lib.semz_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]
    lib.semz_profile_reset.argtypes = [vp]
    lib.semz_profile_reset.restype = None
    _lib = lib
    return lib


class EngineError(RuntimeError):
    """API misuse or a HIP failure (not a data error)."""


class Engine:
    """One context on one GPU (semz_open), or on several (semz_open_multi: `device` a list whose first entry is
    the primary device; the stream calls then spread every predicted batch over all of them)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*device)
            rc = self.lib.semz_open_multi(devs, len(device), C.byref(h))
        else:
            rc = self.lib.semz_open(device, C.byref(h))
        if rc != 0:
            raise EngineError(f"semz_open(device={device}) failed with {rc}: no usable MI355X; "
                              "this library has no CPU fallback")
        self.h = h

    @property
    def device_count(self):
        return self.lib.semz_device_count(self.h)

    def close(self):
        if self.h:
            self.lib.semz_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(f"rc={rc}: {self.lib.semz_last_error(self.h).decode()}")

    def set_option(self, name, value):
        self._chk(self.lib.semz_set_option(self.h, name.encode(), value))

    def get_option(self, name):
        v = C.c_long()
        self._chk(self.lib.semz_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def decode_batch(self, data, units, out_cap):
        """Host-buffer batch decode. Returns (bytes-like output buffer, list of Result)."""
        n = len(units)
        uarr = (Unit * n)(*units)
        res = (Result * n)()
        out = C.create_string_buffer(max(out_cap, 1))
        self._chk(self.lib.semz_decode_batch(self.h, bytes(data), len(data), uarr, n,
                                             C.cast(out, C.c_void_p), out_cap, res))
        return out, list(res)

    # ---- device-resident API used by bench.py and the large-size tests ----
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.semz_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        self._chk(self.lib.semz_dev_free(self.h, p))

    def upload(self, dptr, buf, nbytes=None, offset=0):
        nbytes = len(buf) if nbytes is None else nbytes
        src = (C.c_char * nbytes).from_buffer_copy(buf) if isinstance(buf, (bytes, bytearray, memoryview)) else buf
        self._chk(self.lib.semz_dev_upload(self.h, C.c_void_p(dptr.value + offset), src, nbytes))

    def download(self, dptr, nbytes, offset=0):
        out = C.create_string_buffer(max(nbytes, 1))
        self._chk(self.lib.semz_dev_download(self.h, out, C.c_void_p(dptr.value + offset), nbytes))
        return out.raw[:nbytes]

    def decode_batch_device(self, kind, d_in, in_len, d_units, nunits, d_out, out_cap, d_results):
        self._chk(self.lib.semz_decode_batch_device(self.h, kind, d_in, in_len, d_units, nunits,
                                                    d_out, out_cap, d_results))

    def decode_batch_device_noalloc(self, kind, d_in, in_len, d_units, nunits, d_out, out_cap, d_results):
        self._chk(self.lib.semz_decode_batch_device_noalloc(self.h, kind, d_in, in_len, d_units, nunits,
                                                            d_out, out_cap, d_results))

    def sync(self):
        self._chk(self.lib.semz_sync(self.h))

    def event(self):
        e = C.c_void_p()
        self._chk(self.lib.semz_event_create(self.h, C.byref(e)))
        return e

    def record(self, ev):
        self._chk(self.lib.semz_event_record(self.h, ev))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        self._chk(self.lib.semz_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value

    def profile(self, reset=True):
        """[(kernel name, ms)] for the calls since the last reset ("profile" option on)."""
        out = []
        for i in range(self.lib.semz_profile_count(self.h)):
            name, ms = C.c_char_p(), C.c_float()
            self._chk(self.lib.semz_profile_get(self.h, i, C.byref(name), C.byref(ms)))
            out.append((name.value.decode(), ms.value))
        if reset:
            self.lib.semz_profile_reset(self.h)
        return out

    def stream_decode(self, fmt, data, all_members=True):
        return Decoded(self, fmt, data, all_members)

    def decode_stream_into(self, fmt, data, out_buf, all_members=True):
        """decompress-all into a caller-owned ctypes buffer. Returns (rc, out_len, Result, StreamInfo); rc 1 = too small."""
        out_len = C.c_size_t()
        res = Result()
        info = StreamInfo()
        rc = self.lib.semz_decode_stream_into(self.h, fmt, bytes(data), len(data), 1 if all_members else 0,
                                              C.cast(out_buf, C.c_void_p), C.sizeof(out_buf), C.byref(out_len),
                                              C.byref(res), C.byref(info))
        if rc not in (0, 1):
            self._chk(rc)
        return rc, out_len.value, res, info

    def decode_stream(self, fmt, data, all_members=True):
        """decompress / decompress-all on the GPU. Returns (bytes, Result, StreamInfo)."""
        out = C.c_void_p()
        out_len = C.c_size_t()
        res = Result()
        info = StreamInfo()
        self._chk(self.lib.semz_decode_stream(self.h, fmt, bytes(data), len(data), 1 if all_members else 0,
                                              C.byref(out), C.byref(out_len), C.byref(res), C.byref(info)))
        try:
            payload = C.string_at(out, out_len.value) if out.value else b""
        finally:
            if out.value:
                self.lib.semz_free(out)
        return payload, res, info


class Decoded:
    """A decoded stream held in device memory (semz_stream_decode): members, ranged reads, release."""

    def __init__(self, engine, fmt, data, all_members=True):
        self.engine = engine
        self.res = Result()
        self.info = StreamInfo()
        h = C.c_void_p()
        engine._chk(engine.lib.semz_stream_decode(engine.h, fmt, bytes(data), len(data), 1 if all_members else 0,
                                                  C.byref(h), C.byref(self.res), C.byref(self.info)))
        self.h = h
        m = C.POINTER(Member)()
        n = C.c_size_t()
        engine._chk(engine.lib.semz_stream_members(self.h, C.byref(m), C.byref(n)))
        self.members = [(m[i].in_end, m[i].out_end) for i in range(n.value)]
        self.out_len = self.res.out_len

    def read(self, offset, length):
        buf = C.create_string_buffer(max(length, 1))
        self.engine._chk(self.engine.lib.semz_stream_read(self.h, offset, buf, length))
        return buf.raw[:length]

    def release(self):
        if self.h:
            self.engine.lib.semz_stream_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def strerror(status):
    return load().semz_strerror(status).decode()
