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


def arm64_encode(data, start=0):
    """Forward ARM64 BCJ filter (XZ Utils 5.4 src/liblzma/simple/arm64.c with is_encoder set): liblzma 5.2.5
    and CPython's lzma have no ARM64 encoder, so the test corpus for the decoder is made here."""
    b = bytearray(data)
    for i in range(0, len(b) - 3, 4):
        instr = int.from_bytes(b[i:i + 4], "little")
        pc = (start + i) & 0xFFFFFFFF
        if instr >> 26 == 0x25:
            instr = 0x94000000 | ((instr + (pc >> 2)) & 0x03FFFFFF)
        elif instr & 0x9F000000 == 0x90000000:
            src = ((instr >> 29) & 3) | ((instr >> 3) & 0x001FFFFC)
            if (src + 0x00020000) & 0x001C0000:
                continue
            dest = (src + (pc >> 12)) & 0xFFFFFFFF
            instr = (instr & 0x9000001F) | ((dest & 3) << 29) | ((dest & 0x0003FFFC) << 3) | ((0 - (dest & 0x00020000)) & 0x00E00000)
        else:
            continue
        b[i:i + 4] = instr.to_bytes(4, "little")
    return bytes(b)


def xz_single_block(filtered, plain, filter_id, crc64):
    """A one-block .xz stream by hand: `filtered` is LZMA2-compressed behind the BCJ filter `filter_id`
    (no properties), the CRC64 check is the one of `plain`.  crc64: a function bytes -> int."""
    import zlib

    def varint(v):
        out = bytearray()
        while True:
            out.append(v & 0x7F)
            v >>= 7
            if not v:
                return bytes(out)
            out[-1] |= 0x80

    def with_crc(body):
        return body + zlib.crc32(body).to_bytes(4, "little")

    raw = lzma.compress(filtered, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 6, "dict_size": 1 << 20}])
    flags = b"\x00\x04"                                                  # CRC64
    header = b"\xfd7zXZ\x00" + with_crc(flags)
    bh = bytes([2, 0x01, filter_id, 0x00, 0x21, 0x01, 16, 0x00])         # 12-byte header, two filters, 1 MiB dictionary
    block = with_crc(bh) + raw + b"\0" * (-len(raw) % 4) + crc64(plain).to_bytes(8, "little")
    index = b"\x00" + varint(1) + varint(12 + len(raw) + 8) + varint(len(plain))
    index = with_crc(index + b"\0" * (-len(index) % 4))
    footer = ((len(index) // 4) - 1).to_bytes(4, "little") + flags
    return header + block + index + zlib.crc32(footer).to_bytes(4, "little") + footer + b"YZ"


def varint(v):
    out = bytearray()
    while True:
        out.append(v & 0x7F)
        v >>= 7
        if not v:
            return bytes(out)
        out[-1] |= 0x80


def xz_rewrite_index(data, records=None, edit=None):
    """A single-stream .xz with its index replaced: `records` = [(unpadded size, uncompressed size)], or
    `edit` = a function old records -> new records.  Index CRC, backward size and footer CRC are made
    consistent, so only the forward walk (xz.lisp:186-203) can tell that the index lies."""
    import zlib
    footer = data[-12:]
    assert footer[-2:] == b"YZ"
    bsize = 4 * (int.from_bytes(footer[4:8], "little") + 1)
    idx0 = len(data) - 12 - bsize
    if records is None:
        p = idx0 + 1
        def rd():
            nonlocal p
            v, sh = 0, 0
            while True:
                b = data[p]; p += 1
                v |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    return v
        old = [(rd(), rd()) for _ in range(rd())]
        records = edit(old)
    index = b"\x00" + varint(len(records)) + b"".join(varint(a) + varint(b) for a, b in records)
    index += b"\0" * (-len(index) % 4)
    index += zlib.crc32(index).to_bytes(4, "little")
    tail = ((len(index) // 4) - 1).to_bytes(4, "little") + footer[8:10]
    return data[:idx0] + index + zlib.crc32(tail).to_bytes(4, "little") + tail + b"YZ"


def xz_block_with_declared_sizes(plain, csize=None, usize=None, check=b"\x00\x01"):
    """A one-block .xz stream whose block header declares the given compressed / uncompressed sizes
    (None = field absent), whatever the block really holds.  CRC32 check by default."""
    import zlib

    def with_crc(body):
        return body + zlib.crc32(body).to_bytes(4, "little")

    raw = lzma.compress(plain, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 6, "dict_size": 1 << 20}])
    flags = (0x40 if csize is not None else 0) | (0x80 if usize is not None else 0)
    body = bytes([flags]) + (varint(csize) if csize is not None else b"") + (varint(usize) if usize is not None else b"")
    body += bytes([0x21, 0x01, 16])
    total = (len(body) + 1 + 3) & ~3
    bh = bytes([total // 4]) + body + b"\0" * (total - 1 - len(body))
    chk = zlib.crc32(plain).to_bytes(4, "little")
    block = with_crc(bh) + raw + b"\0" * (-len(raw) % 4) + chk
    header = b"\xfd7zXZ\x00" + with_crc(check)
    index = b"\x00" + varint(1) + varint(len(bh) + 4 + len(raw) + 4) + varint(len(plain))
    index = with_crc(index + b"\0" * (-len(index) % 4))
    footer = ((len(index) // 4) - 1).to_bytes(4, "little") + check
    return header + block + index + zlib.crc32(footer).to_bytes(4, "little") + footer + b"YZ"
