"""Hand-builders for compressed streams, bit by bit (test infrastructure, nothing here ships).

The system encoders (zlib, libbz2, liblzma) only write well-formed streams of the shapes they prefer.
The reject paths of the reference (README.org:161-354) and the shapes no current encoder writes
(bzip2 randomised blocks, deflate blocks with a single distance code, LZMA2 chunk sequences that change
properties mid-stream) need streams made by hand; these builders follow the published formats
(RFC 1951 / 1950 / 1952, the bzip2 format as implemented by libbz2, Pavlov's lzma-specification.txt,
the .xz file format 1.0.4).
"""
import zlib


# ---------------------------------------------------------------- bit writers

class LsbWriter:
    """Deflate bit order: the first bit goes to bit 0 of the first byte (bits.lisp:118)."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, value, count):
        self.acc |= (value & ((1 << count) - 1)) << self.n
        self.n += count
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8
        return self

    def code(self, code, length):
        """A Huffman code: its bits go out most significant first (RFC 1951 3.1.1)."""
        for i in range(length - 1, -1, -1):
            self.bits((code >> i) & 1, 1)
        return self

    def align(self):
        if self.n:
            self.bits(0, 8 - self.n)
        return self

    def raw(self, data):
        assert self.n == 0
        self.out += data
        return self

    def done(self):
        self.align()
        return bytes(self.out)


class MsbWriter:
    """bzip2 bit order: the first bit goes to bit 7 of the first byte (bits.lisp:121)."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, value, count):
        for i in range(count - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                self.acc, self.n = 0, 0
        return self

    def done(self):
        while self.n:
            self.bits(0, 1)
        return bytes(self.out)


def canonical_codes(lengths):
    """Canonical Huffman codes for a list of code lengths (0 = unused), RFC 1951 3.2.2."""
    maxlen = max(lengths) if lengths else 0
    count = [0] * (maxlen + 2)
    for l in lengths:
        count[l] += 1
    count[0] = 0
    code, nxt = 0, [0] * (maxlen + 2)
    for b in range(1, maxlen + 1):
        code = (code + count[b - 1]) << 1
        nxt[b] = code
    codes = []
    for l in lengths:
        if l:
            codes.append(nxt[l])
            nxt[l] += 1
        else:
            codes.append(None)
    return codes


def complete_lengths(n, short=None):
    """n code lengths whose Kraft sum is exactly 1: 2^(k+1) - n codes of length k, the others k + 1."""
    assert n >= 2
    k = n.bit_length() - 1
    if n == 1 << k:
        return [k] * n
    nshort = (1 << (k + 1)) - n
    return [k] * nshort + [k + 1] * (n - nshort)


# ---------------------------------------------------------------- deflate

CLEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]     # deflate.lisp:174
# a complete code-length code over all 19 symbols: thirteen codes of four bits, six of five
CLEN_LENGTHS = [4] * 13 + [5] * 6

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class DeflateBlock:
    """One Huffman block under construction: the caller picks the code lengths and emits tokens."""

    def __init__(self, w, lit_lengths, dist_lengths):
        self.w = w
        self.lit_lengths, self.dist_lengths = lit_lengths, dist_lengths
        self.lit_codes, self.dist_codes = canonical_codes(lit_lengths), canonical_codes(dist_lengths)

    def sym(self, s):
        self.w.code(self.lit_codes[s], self.lit_lengths[s])
        return self

    def lit(self, data):
        for b in data:
            self.sym(b)
        return self

    def length(self, ln):
        i = max(k for k in range(29) if LEN_BASE[k] <= ln and (k == 28 or ln != 258))
        self.sym(257 + i)
        self.w.bits(ln - LEN_BASE[i], LEN_EXTRA[i])
        return self

    def dist_sym(self, c, extra=0, nextra=0):
        self.w.code(self.dist_codes[c], self.dist_lengths[c])
        self.w.bits(extra, nextra)
        return self

    def dist(self, d):
        c = max(k for k in range(30) if DIST_BASE[k] <= d)
        return self.dist_sym(c, d - DIST_BASE[c], DIST_EXTRA[c])

    def match(self, ln, d):
        return self.length(ln).dist(d)

    def eob(self):
        return self.sym(256)


def fixed_block(w, final=True):
    """BFINAL, BTYPE 01 and the fixed trees (deflate.lisp:146-158): 288 lit/len codes, 32 distance codes."""
    w.bits(1 if final else 0, 1).bits(1, 2)
    return DeflateBlock(w, [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, [5] * 32)


def dynamic_block(w, lit_lengths, dist_lengths, final=True, clen_lengths=None, raw_codelens=None, hclen=19):
    """BFINAL, BTYPE 10 and a dynamic header (deflate.lisp:160-201).  The code lengths are sent one by one with
    the complete code CLEN_LENGTHS unless raw_codelens gives the exact sequence of (symbol, extra value) pairs."""
    clen_lengths = CLEN_LENGTHS if clen_lengths is None else clen_lengths
    w.bits(1 if final else 0, 1).bits(2, 2)
    w.bits(len(lit_lengths) - 257, 5).bits(len(dist_lengths) - 1, 5).bits(hclen - 4, 4)
    for s in CLEN_ORDER[:hclen]:
        w.bits(clen_lengths[s], 3)
    cc = canonical_codes(clen_lengths)
    seq = raw_codelens if raw_codelens is not None else [(l, None) for l in list(lit_lengths) + list(dist_lengths)]
    for s, extra in seq:
        w.code(cc[s], clen_lengths[s])
        if s == 16:
            w.bits(extra, 2)
        elif s == 17:
            w.bits(extra, 3)
        elif s == 18:
            w.bits(extra, 7)
    return DeflateBlock(w, list(lit_lengths), list(dist_lengths))


def stored_block(w, data, final=True, nlen=None):
    w.bits(1 if final else 0, 1).bits(0, 2).align()
    n = len(data)
    w.bits(n, 16).bits((n ^ 0xFFFF) if nlen is None else nlen, 16)
    return w.raw(data)


# complete lit/len code-length sets (Kraft sum exactly 1, huffman.lisp:166-184)
LIT257 = [8] * 255 + [9] * 2                 # literals + end of block
LIT286 = [8] * 226 + [9] * 60                # + the 29 length codes
LIT288 = [8] * 224 + [9] * 64                # + the two codes that must never occur (README.org:137-141)
DIST30 = [4] * 2 + [5] * 28
DIST32 = [5] * 32                            # + distance codes 30 / 31


def zlib_wrap(raw, data, cmf=0x78, flg=None, adler=None, dictid=None):
    if flg is None:
        flg = 0x20 if dictid is not None else 0
        flg |= (31 - (cmf * 256 + flg) % 31) % 31
    out = bytes([cmf, flg])
    if dictid is not None:
        out += dictid.to_bytes(4, "big")
    return out + raw + (zlib.adler32(data) if adler is None else adler).to_bytes(4, "big")


def gzip_wrap(raw, data, flg=0, extra=b"", name=b"", comment=b"", hcrc=None, magic=b"\x1f\x8b", cm=8, crc=None, isize=None):
    h = magic + bytes([cm, flg]) + b"\0\0\0\0\0\3"
    if flg & 4:
        h += len(extra).to_bytes(2, "little") + extra
    if flg & 8:
        h += name + b"\0"
    if flg & 16:
        h += comment + b"\0"
    if flg & 2:
        h += ((zlib.crc32(h) & 0xFFFF) if hcrc is None else hcrc).to_bytes(2, "little")
    crc = zlib.crc32(data) if crc is None else crc
    isize = len(data) if isize is None else isize
    return h + raw + crc.to_bytes(4, "little") + (isize & 0xFFFFFFFF).to_bytes(4, "little")


# ---------------------------------------------------------------- bzip2

BZ_RAND = None      # filled by bz_rand_table() from the oracle's copy check in the tests (the table is data, not code)


def bz_crc(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc ^ 0xFFFFFFFF


def bz_rle1(data):
    """bzip2's first run-length stage: four equal bytes, then a count byte 0..251 of further repeats."""
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        j = i
        while j < n and data[j] == data[i] and j - i < 255:
            j += 1
        run = j - i
        if run >= 4:
            out += bytes([data[i]]) * 4 + bytes([run - 4])
        else:
            out += bytes([data[i]]) * run
        i = j
    return bytes(out)


def bz_bwt(block):
    """Burrows-Wheeler transform by sorting all rotations (small blocks only). Returns (last column, origPtr)."""
    n = len(block)
    doubled = block + block
    order = sorted(range(n), key=lambda i: doubled[i:i + n])
    last = bytes(doubled[i + n - 1] for i in order)
    return last, order.index(0)


def bz_symbols(last, used):
    """MTF + RUNA/RUNB (zero-run) coding of the BWT output over the sorted list of used byte values."""
    mtf = list(used)
    syms, run = [], 0

    def flush():
        nonlocal run
        while run:
            run -= 1
            syms.append(run & 1)            # RUNA = 0 (digit 1), RUNB = 1 (digit 2), bijective base 2
            run >>= 1

    for b in last:
        k = mtf.index(b)
        if k == 0:
            run += 1
            continue
        flush()
        syms.append(k + 1)
        mtf.insert(0, mtf.pop(k))
    flush()
    syms.append(len(used) + 1)              # end of block
    return syms


def bz_rand_positions(n, table):
    """Positions the randomisation flips (derandomize-bzip2-block, bzip2.lisp:112-121)."""
    pos, out, i = table[0] - 2, [], 0
    while pos < n:
        out.append(pos)
        i = (i + 1) % 512
        pos += table[i]
    return out


def bz_block(w, data=None, *, block=None, randomised=False, rand_table=None, crc=None, orig=None, l1_map=None,
             ntrees=2, ntrees_field=None, nsel_field=None, selector_codes=None, tree_start=None, tree_lengths=None,
             tree_delta_bits=None, symbols=None, used=None, magic=0x314159265359):
    """One bzip2 block into MsbWriter w.  `data` is what the block decodes to; every field of the block can be
    overridden to provoke a reject path.  Returns the block CRC written."""
    if block is None:
        block = bz_rle1(data)
    pre = bytearray(block)
    if randomised:
        for p in bz_rand_positions(len(pre), rand_table):
            pre[p] ^= 1
    if used is None:
        used = sorted(set(pre))
    if symbols is None:
        last, optr = bz_bwt(bytes(pre))
        symbols = bz_symbols(last, used)
    else:
        optr = 0
    crc_val = bz_crc(data if data is not None else b"") if crc is None else crc
    w.bits(magic, 48).bits(crc_val, 32).bits(1 if randomised else 0, 1).bits(optr if orig is None else orig, 24)
    l1 = 0
    for b in used:
        l1 |= 1 << (15 - (b >> 4))
    w.bits(l1 if l1_map is None else l1_map, 16)
    for i in range(16):
        if (l1 >> (15 - i)) & 1:
            l2 = 0
            for b in used:
                if b >> 4 == i:
                    l2 |= 1 << (15 - (b & 15))
            w.bits(l2, 16)
    alpha = len(used) + 2
    nsel = (len(symbols) + 49) // 50
    w.bits(ntrees if ntrees_field is None else ntrees_field, 3)
    w.bits(nsel if nsel_field is None else nsel_field, 15)
    if selector_codes is None:
        selector_codes = [(0, 1)] * (nsel if nsel_field is None else nsel_field)       # MTF index 0 = "0"
    for v, nb in selector_codes:
        w.bits(v, nb)
    lengths = complete_lengths(alpha) if tree_lengths is None else tree_lengths
    for _ in range(ntrees):
        if tree_delta_bits is not None:
            w.bits(lengths[0] if tree_start is None else tree_start, 5)
            for v, nb in tree_delta_bits:
                w.bits(v, nb)
            continue
        cur = lengths[0] if tree_start is None else tree_start
        w.bits(cur, 5)
        for l in lengths:
            while cur < l:
                w.bits(2, 2)
                cur += 1
            while cur > l:
                w.bits(3, 2)
                cur -= 1
            w.bits(0, 1)
    codes = canonical_codes(lengths)
    for s in symbols:
        w.bits(codes[s], lengths[s])
    return crc_val


def bz_stream(blocks, level=1, header=None, stream_crc=None, footer=0x177245385090):
    """'BZh' + level, the blocks (callables taking the writer and returning their CRC), footer magic, stream CRC."""
    w = MsbWriter()
    hdr = b"BZh" + bytes([0x30 + level]) if header is None else header
    for b in hdr:
        w.bits(b, 8)
    combined = 0
    for blk in blocks:
        c = blk(w)
        combined = (((combined << 1) | (combined >> 31)) & 0xFFFFFFFF) ^ c
    if footer is not None:
        w.bits(footer, 48).bits(combined if stream_crc is None else stream_crc, 32)
    return w.done()


# ---------------------------------------------------------------- LZMA

class RangeEncoder:
    def __init__(self):
        self.low, self.range, self.cache, self.cache_size = 0, 0xFFFFFFFF, 0, 1
        self.out = bytearray()

    def shift_low(self):
        if self.low < 0xFF000000 or self.low >= 1 << 32:
            carry, temp = self.low >> 32, self.cache
            while True:
                self.out.append((temp + carry) & 0xFF)
                temp = 0xFF
                self.cache_size -= 1
                if not self.cache_size:
                    break
            self.cache = (self.low >> 24) & 0xFF
        self.cache_size += 1
        self.low = (self.low & 0x00FFFFFF) << 8

    def normalize(self):
        while self.range < 1 << 24:
            self.range = (self.range << 8) & 0xFFFFFFFF
            self.shift_low()

    def bit(self, probs, key, b):
        p = probs.get(key, 1024)
        bound = (self.range >> 11) * p
        if b == 0:
            self.range = bound
            probs[key] = p + ((2048 - p) >> 5)
        else:
            self.low += bound
            self.range -= bound
            probs[key] = p - (p >> 5)
        self.normalize()

    def direct(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.range >>= 1
            if (value >> i) & 1:
                self.low += self.range
            self.normalize()

    def finish(self, top=False):
        """Flushes; with top=True the bytes written place the decoder's `code` at range - 1, the last value of
        the current interval (lzma.lisp:186-188 can only fail from there)."""
        if top:
            self.low += self.range - 1
        for _ in range(5):
            self.shift_low()
        return bytes(self.out)


class LzmaEncoder:
    """Symbol-level LZMA encoder: the caller names every symbol (lzma.lisp:708-765), nothing is searched."""

    def __init__(self, lc=3, lp=0, pb=2, history=b""):
        self.lc, self.lp, self.pb = lc, lp, pb
        self.rc = RangeEncoder()
        self.probs = {}
        self.state = 0
        self.reps = [0, 0, 0, 0]
        self.out = bytearray(history)      # dictionary contents (needed for literal contexts and match bytes)
        self.total = 0                     # total-i of the reference: position used for pos-state / literal context

    # -- state carried across LZMA2 chunks
    def reset_state(self, lc=None, lp=None, pb=None):
        if lc is not None:
            self.lc, self.lp, self.pb = lc, lp, pb
        self.probs, self.state, self.reps, self.total = {}, 0, [0, 0, 0, 0], 0

    def new_chunk(self):
        self.rc = RangeEncoder()

    def _ps(self):
        return self.total & ((1 << self.pb) - 1)

    def _tree(self, key, nbits, value):
        m = 1
        for i in range(nbits - 1, -1, -1):
            b = (value >> i) & 1
            self.rc.bit(self.probs, (key, m), b)
            m = (m << 1) | b

    def _rtree(self, key, nbits, value):
        m = 1
        for i in range(nbits):
            b = (value >> i) & 1
            self.rc.bit(self.probs, (key, m), b)
            m = (m << 1) | b

    def _length(self, kind, ln, ps):
        ln -= 2
        if ln < 8:
            self.rc.bit(self.probs, (kind, "c0"), 0)
            self._tree((kind, "low", ps), 3, ln)
        elif ln < 16:
            self.rc.bit(self.probs, (kind, "c0"), 1)
            self.rc.bit(self.probs, (kind, "c1"), 0)
            self._tree((kind, "mid", ps), 3, ln - 8)
        else:
            self.rc.bit(self.probs, (kind, "c0"), 1)
            self.rc.bit(self.probs, (kind, "c1"), 1)
            self._tree((kind, "high"), 8, ln - 16)

    def _distance(self, dist, ln, stop_before_direct=False):
        ls = min(ln - 2, 3)
        if dist < 4:
            slot = dist
        else:
            hb = dist.bit_length() - 1
            slot = 2 * hb + ((dist >> (hb - 1)) & 1)
        self._tree(("slot", ls), 6, slot)
        if slot >= 4:
            fb = (slot >> 1) - 1
            base = (2 | (slot & 1)) << fb
            red = dist - base
            if slot < 14:
                self._rtree(("spec", base - slot), fb, red)
            else:
                if stop_before_direct:
                    return
                self.rc.direct(red >> 4, fb - 4)
                self._rtree(("align",), 4, red & 15)

    def _copy(self, dist0, ln):
        for _ in range(ln):
            src = len(self.out) - dist0 - 1
            self.out.append(self.out[src] if 0 <= src < len(self.out) else 0)
        self.total += ln

    def literal(self, byte):
        ps = self._ps()
        self.rc.bit(self.probs, ("islit", self.state, ps), 0)
        prev = self.out[-1] if self.out else 0
        ctx = ((self.total & ((1 << self.lp) - 1)) << self.lc) | (prev >> (8 - self.lc))
        sym = 1
        if self.state >= 7:
            src = len(self.out) - self.reps[0] - 1
            mb = self.out[src] if 0 <= src < len(self.out) else 0
            for i in range(7, -1, -1):
                m, b = (mb >> i) & 1, (byte >> i) & 1
                self.rc.bit(self.probs, ("lit", ctx, 0x100 + (m << 8) + sym), b)
                sym = (sym << 1) | b
                if m != b:
                    break
        while sym < 0x100:
            b = (byte >> (7 - (sym.bit_length() - 1))) & 1
            self.rc.bit(self.probs, ("lit", ctx, sym), b)
            sym = (sym << 1) | b
        self.state = 0 if self.state < 4 else self.state - 3 if self.state < 10 else self.state - 6
        self.out.append(byte)
        self.total += 1
        return self

    def literals(self, data):
        for b in data:
            self.literal(b)
        return self

    def match(self, ln, dist0, stop_before_direct=False):
        """A simple match of length ln (2..273) at distance dist0 + 1; dist0 = 0xFFFFFFFF is the end marker."""
        ps = self._ps()
        self.rc.bit(self.probs, ("islit", self.state, ps), 1)
        self.rc.bit(self.probs, ("issimple", self.state), 0)
        self._length("len", ln, ps)
        self._distance(dist0, ln, stop_before_direct)
        self.reps = [dist0] + self.reps[:3]
        self.state = 7 if self.state < 7 else 10
        if dist0 != 0xFFFFFFFF and not stop_before_direct:
            self._copy(dist0, ln)
        return self

    def marker(self):
        return self.match(2, 0xFFFFFFFF)

    def shortrep(self):
        ps = self._ps()
        self.rc.bit(self.probs, ("islit", self.state, ps), 1)
        self.rc.bit(self.probs, ("issimple", self.state), 1)
        self.rc.bit(self.probs, ("rep0", self.state), 0)
        self.rc.bit(self.probs, ("short", self.state, ps), 0)
        self.state = 9 if self.state < 7 else 11
        self._copy(self.reps[0], 1)
        return self

    def rep(self, idx, ln):
        ps = self._ps()
        self.rc.bit(self.probs, ("islit", self.state, ps), 1)
        self.rc.bit(self.probs, ("issimple", self.state), 1)
        if idx == 0:
            self.rc.bit(self.probs, ("rep0", self.state), 0)
            self.rc.bit(self.probs, ("short", self.state, ps), 1)
        else:
            self.rc.bit(self.probs, ("rep0", self.state), 1)
            if idx == 1:
                self.rc.bit(self.probs, ("rep1", self.state), 0)
            else:
                self.rc.bit(self.probs, ("rep1", self.state), 1)
                self.rc.bit(self.probs, ("rep2", self.state), idx - 2)
            d = self.reps.pop(idx)
            self.reps.insert(0, d)
        self._length("replen", ln, ps)
        self.state = 8 if self.state < 7 else 11
        self._copy(self.reps[0], ln)
        return self

    def finish(self, top=False):
        return self.rc.finish(top)


def lzma_props(lc, lp, pb):
    return (pb * 5 + lp) * 9 + lc


def lzma2_lzma_chunk(payload, usize, control=0xE0, props=None, csize=None):
    """An LZMA chunk: control (0x80 | reset level << 5 | usize bits 20..16), sizes minus one, optional props byte."""
    u, c = usize - 1, (len(payload) if csize is None else csize) - 1
    out = bytes([control | ((u >> 16) & 31), (u >> 8) & 0xFF, u & 0xFF, (c >> 8) & 0xFF, c & 0xFF])
    if props is not None:
        out += bytes([props])
    return out + payload


def lzma2_stored_chunk(data, reset_dict=True, control=None):
    u = len(data) - 1
    return bytes([(1 if reset_dict else 2) if control is None else control, (u >> 8) & 0xFF, u & 0xFF]) + data


def lzma_alone_header(lc=3, lp=0, pb=2, dict_size=1 << 16, size=None, props=None):
    p = lzma_props(lc, lp, pb) if props is None else props
    return bytes([p]) + dict_size.to_bytes(4, "little") + ((1 << 64) - 1 if size is None else size).to_bytes(8, "little")


# ---------------------------------------------------------------- xz

def xz_varint(v):
    out = bytearray()
    while True:
        out.append(v & 0x7F)
        v >>= 7
        if not v:
            return bytes(out)
        out[-1] |= 0x80


def crc32le(b):
    return zlib.crc32(b).to_bytes(4, "little")


def xz_stream_header(check=1, magic=b"\xfd7zXZ\x00", flags=None, crc=None):
    f = bytes([0, check]) if flags is None else flags
    return magic + f + (crc32le(f) if crc is None else crc)


def xz_block_header(filters, csize=None, usize=None, flags=None, pad=None, crc=None, size_byte=None, reserved=0):
    """filters: [(id, props bytes)], in file order (LZMA2 last)."""
    fl = (len(filters) - 1) | (0x40 if csize is not None else 0) | (0x80 if usize is not None else 0) | reserved
    body = bytes([fl if flags is None else flags])
    if csize is not None:
        body += xz_varint(csize) if isinstance(csize, int) else csize
    if usize is not None:
        body += xz_varint(usize) if isinstance(usize, int) else usize
    for fid, props in filters:
        body += xz_varint(fid) + xz_varint(len(props)) + props
    total = (len(body) + 1 + 3) & ~3
    padding = b"\0" * (total - 1 - len(body)) if pad is None else pad
    hdr = bytes([total // 4 if size_byte is None else size_byte]) + body + padding
    return hdr + (crc32le(hdr) if crc is None else crc)


def xz_index(records, pad=None, crc=None, count=None, raw_records=None):
    body = b"\x00" + xz_varint(len(records) if count is None else count)
    body += raw_records if raw_records is not None else b"".join(xz_varint(a) + xz_varint(b) for a, b in records)
    body += (b"\0" * (-len(body) % 4)) if pad is None else pad
    return body + (crc32le(body) if crc is None else crc)


def xz_footer(index_len, check=1, flags=None, crc=None, magic=b"YZ", backward=None):
    f = bytes([0, check]) if flags is None else flags
    tail = ((index_len // 4 - 1) if backward is None else backward).to_bytes(4, "little") + f
    return (crc32le(tail) if crc is None else crc) + tail + magic


def xz_block(header, payload, check_bytes, pad=None):
    return header + payload + ((b"\0" * (-len(payload) % 4)) if pad is None else pad) + check_bytes


def xz_file(plain, lzma2_payload, check=1, filters=None, check_bytes=None, **parts):
    """A one-block .xz around a raw LZMA2 payload; keyword `parts` replace whole sections:
    header=, block_header=, block_pad=, index=, footer=."""
    filters = [(0x21, bytes([16]))] if filters is None else filters
    if check_bytes is None:
        check_bytes = {0: b"", 1: crc32le(plain)}[check]
    bh = parts.get("block_header") or xz_block_header(filters)
    block = xz_block(bh, lzma2_payload, check_bytes, parts.get("block_pad"))
    unpadded = len(bh) + len(lzma2_payload) + len(check_bytes)
    index = parts.get("index") or xz_index([(unpadded, len(plain))])
    return (parts.get("header") or xz_stream_header(check)) + block + index + (parts.get("footer") or xz_footer(len(index), check))
