"""Test helpers: xz / bzip2 / gzip streams made with the system tools and CPython."""
import lzma
import subprocess


def xz_cli(data, *args):
    """xz via the CLI (multi-block streams need --block-size, which CPython's lzma lacks)."""
    return subprocess.run(["xz", "-c", "-T1", *args], input=data, stdout=subprocess.PIPE, check=True).stdout


def xz_py(d, check=lzma.CHECK_CRC64, filters=None):
    if filters is None:
        filters = [{"id": lzma.FILTER_LZMA2, "preset": 6}]
    return lzma.compress(d, lzma.FORMAT_XZ, check, None, filters)


def bgzf(data, block=60000, level=6):
    """BGZF: gzip members that carry their own size in a 'BC' extra subfield."""
    import zlib
    out = bytearray()
    for i in range(0, max(len(data), 1), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 18 + len(body) + 8 - 1
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + (6).to_bytes(2, "little") + b"BC\x02\x00" + bsize.to_bytes(2, "little")
        out += body + zlib.crc32(chunk).to_bytes(4, "little") + (len(chunk) & 0xFFFFFFFF).to_bytes(4, "little")
    return bytes(out)
