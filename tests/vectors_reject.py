"""One hand-built stream per `die` site of the reference on the hot path (include/semz_status.h), following the
edge-case tables of README.org:161-354, plus the accepted oddities those tables list (they pin the boundary
from the other side).  Built with tests/streamkit.py; nothing here is random.

Every vector is a dict:
    name    what it provokes
    route   "stream"  decompress / decompress-all of a whole input:  fmt, all (all-members-p), opt
            "unit"    one unit through the batch entry point:          kind + unit fields (see run_unit)
    data    the input bytes
    status  the status the reference's rules give (include/semz_status.h); 0 = accepted
    detail  (optional) the first argument of the reference's message
    out     (optional) expected output bytes for accepted streams

tests/test_reject_vectors.py pins oracle == status for every vector on the CPU;
tests/test_gpu_reject_vectors.py pins GPU == oracle on status and detail and checks which statuses were seen.
"""
import lzma
import os
import re
import zlib

import streamkit as k

HERE = os.path.dirname(os.path.abspath(__file__))

# the 512-entry randomisation table is data (bzip2's randtable.c, tables.lisp:85-149); the oracle carries it
_src = open(os.path.join(HERE, "..", "oracle", "bzip2.c")).read()
RAND_TABLE = [int(x) for x in re.findall(r"\d+", re.search(r"RAND_TAB\[512\]\s*=\s*\{([^}]*)\}", _src).group(1))]
assert len(RAND_TABLE) == 512 and RAND_TABLE[:3] == [619, 720, 127]

FMT_DEFLATE, FMT_ZLIB, FMT_GZIP, FMT_BZIP2, FMT_RAW_LZMA, FMT_LZMA, FMT_RAW_LZMA2, FMT_LZMA2, FMT_XZ = range(9)

VECTORS = []


def stream(name, fmt, data, status, detail=None, out=None, all=True, opt=0):
    VECTORS.append(dict(name=name, route="stream", fmt=fmt, data=bytes(data), status=status, detail=detail, out=out,
                        all=all, opt=opt))


def unit(name, kind, data, status, detail=None, out=None, **fields):
    VECTORS.append(dict(name=name, route="unit", kind=kind, data=bytes(data), status=status, detail=detail, out=out,
                        fields=fields))


# =============================================================== Deflate (README.org:161-194, deflate.lisp, huffman.lisp)

def deflate_vectors():
    def dyn(lit, dist, body, **kw):
        w = k.LsbWriter()
        b = k.dynamic_block(w, lit, dist, **kw)
        body(b)
        return w.done()

    hello = b"hello, hello, hello!"
    ok = dyn(k.LIT286, k.DIST30, lambda b: b.lit(b"hello, ").match(12, 7).lit(b"!").eob())
    stream("deflate: dynamic block, complete trees", FMT_DEFLATE, ok, 0, out=hello)
    # exact completeness (huffman.lisp:166-184): stricter than zlib
    stream("deflate: lit/len tree underfull", FMT_DEFLATE, dyn([9] * 257, k.DIST30, lambda b: None), 10)
    stream("deflate: lit/len tree overfull", FMT_DEFLATE, dyn([8] * 257, k.DIST30, lambda b: None), 11)
    stream("deflate: a single length-1 lit/len code (zlib accepts, README.org:172-174)", FMT_DEFLATE,
           dyn([0] * 256 + [1], [0], lambda b: None), 10)
    stream("deflate: all lit/len lengths zero", FMT_DEFLATE, dyn([0] * 257, [0], lambda b: None), 10)
    stream("deflate: code-length tree overfull", FMT_DEFLATE,
           dyn(k.LIT257, k.DIST30, lambda b: None, clen_lengths=[4] * 19), 11)
    stream("deflate: code-length tree underfull", FMT_DEFLATE,
           dyn(k.LIT257, k.DIST30, lambda b: None, clen_lengths=[2, 2] + [0] * 17, raw_codelens=[]), 10)
    stream("deflate: distance tree underfull (two codes of length two)", FMT_DEFLATE,
           dyn(k.LIT257, [2, 2], lambda b: None), 10)
    stream("deflate: distance tree overfull", FMT_DEFLATE, dyn(k.LIT257, [1, 1, 1], lambda b: None), 11)
    # block header
    stream("deflate: reserved BTYPE 11", FMT_DEFLATE, k.LsbWriter().bits(1, 1).bits(3, 2).done(), 20)
    w = k.LsbWriter()
    k.stored_block(w, b"abc", nlen=0x1234)
    stream("deflate: stored block LEN / NLEN mismatch", FMT_DEFLATE, w.done(), 21, detail=0x1234)
    w = k.LsbWriter()
    k.stored_block(w, b"", final=False)
    k.stored_block(w, b"xyz")
    stream("deflate: empty stored block, then data (README.org:165-167)", FMT_DEFLATE, w.done(), 0, out=b"xyz")
    # code length sequences (deflate.lisp:176-197)
    stream("deflate: code lengths expand beyond bounds", FMT_DEFLATE,
           dyn(k.LIT257, k.DIST30, lambda b: None, raw_codelens=[(8, None)] * 200 + [(18, 127)]), 22)
    stream("deflate: repeat code 16 with nothing to repeat", FMT_DEFLATE,
           dyn(k.LIT257, k.DIST30, lambda b: None, raw_codelens=[(16, 0)]), 23)
    # 224 x 8, 32 x 9, then thirty-four fives: the last two lit/len lengths and all 32 distance lengths, sent as one
    # length and repeats that run across the lit/len - distance boundary (deflate.lisp:166-169)
    stream("deflate: repeat codes run across the lit/len - distance boundary", FMT_DEFLATE,
           dyn([8] * 224 + [9] * 32 + [5] * 2, [5] * 32, lambda b: b.lit(b"ok").eob(),
               raw_codelens=[(8, None)] + [(16, 3)] * 37 + [(8, None)] + [(9, None)] + [(16, 3)] * 5 + [(9, None)] +
               [(5, None)] + [(16, 3)] * 5 + [(16, 0)]), 0, out=b"ok")
    # token level (deflate.lisp:272-338)
    w = k.LsbWriter()
    k.fixed_block(w).lit(b"ab").length(3).dist_sym(30)
    stream("deflate: fixed block, distance code 30", FMT_DEFLATE, w.done() + b"\0\0", 24, detail=30)
    w = k.LsbWriter()
    k.fixed_block(w).lit(b"ab").length(3).dist_sym(31)
    stream("deflate: fixed block, distance code 31", FMT_DEFLATE, w.done() + b"\0\0", 24, detail=31)
    stream("deflate: dynamic tree defines distance codes 30/31 and uses 30", FMT_DEFLATE,
           dyn(k.LIT286, k.DIST32, lambda b: b.lit(b"ab").length(3).dist_sym(30)) + b"\0\0", 24, detail=30)
    stream("deflate: trees define 286/287 and 30/31 without using them (README.org:137-141)", FMT_DEFLATE,
           dyn(k.LIT288, k.DIST32, lambda b: b.lit(b"abc").match(6, 3).eob()), 0, out=b"abcabcabc")
    w = k.LsbWriter()
    k.fixed_block(w).lit(b"ab").sym(286)
    stream("deflate: fixed block, lit/len code 286", FMT_DEFLATE, w.done() + b"\0\0", 28, detail=286)
    w = k.LsbWriter()
    k.fixed_block(w).lit(b"ab").sym(287)
    stream("deflate: fixed block, lit/len code 287", FMT_DEFLATE, w.done() + b"\0\0", 28, detail=287)
    # distance tree special cases (deflate.lisp:91-122)
    stream("deflate: literals-only block (one distance code of length zero)", FMT_DEFLATE,
           dyn(k.LIT286, [0], lambda b: b.lit(b"only literals").eob()), 0, out=b"only literals")
    stream("deflate: length code in a literals-only block", FMT_DEFLATE,
           dyn(k.LIT286, [0], lambda b: b.lit(b"abc").length(3)) + b"\0\0", 26, detail=257)
    single = lambda bit: dyn(k.LIT286, [0, 0, 1], lambda b: (b.lit(b"abc").length(4), b.w.bits(bit, 1), b.eob()))
    stream("deflate: single distance code, coded as 0", FMT_DEFLATE, single(0), 0, out=b"abcabca")
    stream("deflate: single distance code, coded as 1", FMT_DEFLATE, single(1) + b"\0\0", 25)
    single30 = dyn(k.LIT286, [0] * 30 + [1], lambda b: (b.lit(b"abc").length(4), b.w.bits(0, 1), b.eob()))
    stream("deflate: the single distance code is code 30", FMT_DEFLATE, single30 + b"\0\0", 25)
    # window (deflate.lisp:329-331)
    w = k.LsbWriter()
    k.fixed_block(w).lit(b"abc").match(3, 4).eob()
    stream("deflate: distance reaches before the start of the data", FMT_DEFLATE, w.done(), 27, detail=4)
    w = k.LsbWriter()
    b = k.fixed_block(w)
    b.lit(bytes(300)).match(3, 300).eob()
    unit("deflate: distance inside the data but beyond a 256-byte window", 0, w.done(), 27, detail=300, param0=256, cap=400)
    unit("deflate: the same stream with a 512-byte window", 0, w.done(), 0, out=bytes(303), param0=512, cap=400)
    w = k.LsbWriter()
    k.fixed_block(w).match(5, 2).lit(b"!").eob()
    unit("deflate: match into the preset history (prefix, deflate.lisp:419-425)", 0, w.done(), 0, out=b"ZYZYZ!",
         prefix=b"xxZY", cap=16)
    unit("deflate: match beyond the preset history", 0, w.done(), 27, detail=2, prefix=b"Y", cap=16)
    unit("deflate: prefix longer than the window counts only up to the window", 0, w.done(), 27, detail=2,
         prefix=b"xxZY", param0=1, cap=16)
    # end of input
    stream("deflate: empty input", FMT_DEFLATE, b"", 1)
    stream("deflate: input ends inside a dynamic header", FMT_DEFLATE, ok[:6], 1)
    stream("deflate: input ends inside the tokens", FMT_DEFLATE, ok[:-2], 1)
    stream("deflate: tree without an end-of-block code runs into the end of the input (deflate.lisp:198-200)", FMT_DEFLATE,
           dyn([8] * 256, k.DIST30, lambda b: b.lit(b"no end")), 1)
    w = k.LsbWriter()
    k.stored_block(w, b"abcdef")
    stream("deflate: stored block cut short", FMT_DEFLATE, w.done()[:-2], 1)
    stream("deflate: trailing data in all-members mode (interface.lisp:125-130)", FMT_DEFLATE, ok + b"\1\2\3", 170)
    stream("deflate: trailing data ignored for one member", FMT_DEFLATE, ok + b"\1\2\3", 0, out=hello, all=False)

    # ---- zlib (zlib.lisp:21-78)
    z = k.zlib_wrap(ok, hello)
    stream("zlib: valid member", FMT_ZLIB, z, 0, out=hello)
    stream("zlib: compression method 7", FMT_ZLIB, k.zlib_wrap(ok, hello, cmf=0x77), 40, detail=7)
    stream("zlib: CINFO 8", FMT_ZLIB, k.zlib_wrap(ok, hello, cmf=0x88), 41, detail=8)
    stream("zlib: flag checksum", FMT_ZLIB, k.zlib_wrap(ok, hello, flg=0x9D), 42)
    stream("zlib: FDICT without a dictionary function", FMT_ZLIB, k.zlib_wrap(ok, hello, dictid=0x01020304), 43)
    unit("zlib: dictionary function does not know the id", 1, k.zlib_wrap(ok, hello, dictid=0x01020304), 44,
         detail=0x01020304, flags=1, cap=64)
    stream("zlib: Adler-32 mismatch", FMT_ZLIB, k.zlib_wrap(ok, hello, adler=zlib.adler32(hello) ^ 1), 45,
           detail=zlib.adler32(hello) ^ 1)
    stream("zlib: trailer cut", FMT_ZLIB, z[:-1], 1)
    stream("zlib: header cut", FMT_ZLIB, z[:1], 1)
    w = k.LsbWriter()
    b = k.fixed_block(w)
    b.lit(bytes(300)).match(3, 300).eob()
    d303 = bytes(303)
    stream("zlib: CINFO 0 window (256) enforced (README.org:184)", FMT_ZLIB, k.zlib_wrap(w.done(), d303, cmf=0x08), 27, detail=300)
    stream("zlib: the same tokens under CINFO 1", FMT_ZLIB, k.zlib_wrap(w.done(), d303, cmf=0x18), 0, out=d303)

    # ---- gzip (gzip.lisp:10-115)
    g = k.gzip_wrap(ok, hello)
    stream("gzip: valid member", FMT_GZIP, g, 0, out=hello)
    stream("gzip: magic", FMT_GZIP, k.gzip_wrap(ok, hello, magic=b"\x1f\x8c"), 50)
    stream("gzip: compression method 7", FMT_GZIP, k.gzip_wrap(ok, hello, cm=7), 51, detail=7)
    stream("gzip: reserved flag bit 5", FMT_GZIP, k.gzip_wrap(ok, hello, flg=0x20), 52)
    stream("gzip: extra subfield runs past the field", FMT_GZIP, k.gzip_wrap(ok, hello, flg=4, extra=b"AB\x05\x00xy"), 53)
    stream("gzip: extra field too short for a subfield header", FMT_GZIP, k.gzip_wrap(ok, hello, flg=4, extra=b"AB\x01"), 53)
    stream("gzip: all optional fields, header CRC right", FMT_GZIP,
           k.gzip_wrap(ok, hello, flg=0x1E, extra=b"AB\x02\x00xy", name=b"n", comment=b"c"), 0, out=hello)
    stream("gzip: header CRC wrong", FMT_GZIP, k.gzip_wrap(ok, hello, flg=2, hcrc=0xBEEF), 54, detail=0xBEEF)
    stream("gzip: data CRC wrong", FMT_GZIP, k.gzip_wrap(ok, hello, crc=0x12345678), 55, detail=0x12345678)
    stream("gzip: ISIZE wrong", FMT_GZIP, k.gzip_wrap(ok, hello, isize=21), 56, detail=21)
    stream("gzip: CRC and ISIZE both wrong: the CRC is reported (gzip.lisp:105-113)", FMT_GZIP,
           k.gzip_wrap(ok, hello, crc=1, isize=2), 55, detail=1)
    stream("gzip: two members, four bytes of garbage (too short for a header: eof, gzip.lisp:50)", FMT_GZIP, g + g + b"junk", 1)
    stream("gzip: two members, twelve bytes of garbage", FMT_GZIP, g + g + b"junk junk 12", 50)
    stream("gzip: two members", FMT_GZIP, g + g, 0, out=hello * 2)
    stream("gzip: name not terminated", FMT_GZIP, k.gzip_wrap(ok, hello, flg=8, name=b"abc")[:13], 1)


# =============================================================== bzip2 (README.org:198-227, bzip2.lisp)

def bzip2_vectors():
    data = b"she sells sea shells by the sea shore; " * 12 + bytes(40) + b"z" * 300
    blk = lambda **kw: (lambda w: k.bz_block(w, data, **kw))
    stream("bzip2: valid stream", FMT_BZIP2, k.bz_stream([blk()]), 0, out=data)
    stream("bzip2: two blocks", FMT_BZIP2, k.bz_stream([blk(), lambda w: k.bz_block(w, b"second block")]), 0,
           out=data + b"second block")
    stream("bzip2: randomised block (bzip2.lisp:112-121; no current encoder writes one)", FMT_BZIP2,
           k.bz_stream([blk(randomised=True, rand_table=RAND_TABLE)]), 0, out=data)
    big = bytes((i * 7 + (i >> 5)) & 0xFF for i in range(4000))
    stream("bzip2: randomised block long enough for six flips", FMT_BZIP2,
           k.bz_stream([lambda w: k.bz_block(w, big, randomised=True, rand_table=RAND_TABLE)]), 0, out=big)
    stream("bzip2: six trees", FMT_BZIP2, k.bz_stream([blk(ntrees=6)]), 0, out=data)
    stream("bzip2: stream magic", FMT_BZIP2, k.bz_stream([blk()], header=b"BYh1"), 60, detail=0x4259)
    stream("bzip2: version byte", FMT_BZIP2, k.bz_stream([blk()], header=b"BZg1"), 61, detail=0x67)
    stream("bzip2: level 0", FMT_BZIP2, k.bz_stream([blk()], header=b"BZh0"), 62, detail=0x30)
    stream("bzip2: level ':'", FMT_BZIP2, k.bz_stream([blk()], header=b"BZh:"), 62, detail=0x3A)
    stream("bzip2: block magic", FMT_BZIP2, k.bz_stream([blk(magic=0x314159265358)]), 63)
    stream("bzip2: no used symbols", FMT_BZIP2, k.bz_stream([blk(l1_map=0)]), 64)
    stream("bzip2: one tree", FMT_BZIP2, k.bz_stream([blk(ntrees=1, ntrees_field=1)]), 65, detail=1)
    stream("bzip2: seven trees", FMT_BZIP2, k.bz_stream([blk(ntrees_field=7)]), 65, detail=7)
    stream("bzip2: no selectors", FMT_BZIP2, k.bz_stream([blk(nsel_field=0)]), 66)
    stream("bzip2: selector names a tree that does not exist", FMT_BZIP2,
           k.bz_stream([blk(selector_codes=[(0, 1), (6, 3)] + [(0, 1)] * 40)]), 67, detail=2)
    stream("bzip2: tree start length zero", FMT_BZIP2, k.bz_stream([blk(tree_start=0, tree_delta_bits=[(0, 1)])]), 68)
    stream("bzip2: code length decremented below one", FMT_BZIP2,
           k.bz_stream([blk(tree_start=1, tree_delta_bits=[(3, 2), (0, 1)])]), 68)
    stream("bzip2: code length incremented beyond twenty", FMT_BZIP2,
           k.bz_stream([blk(tree_start=20, tree_delta_bits=[(2, 2), (0, 1)])]), 68)
    stream("bzip2: tree overfull (all codes one bit)", FMT_BZIP2,
           k.bz_stream([blk(tree_lengths=[1] * 40, used=list(range(38)), symbols=[39])]), 11)
    stream("bzip2: tree underfull (libbz2 accepts, README.org:203-206)", FMT_BZIP2,
           k.bz_stream([blk(tree_lengths=[7] * 40, used=list(range(38)), symbols=[39])]), 10)
    # symbols (bzip2.lisp:222-287); alphabet of two bytes: RUNA RUNB 2 EOB
    two = [0x61, 0x62]
    sym = lambda symbols, **kw: (lambda w: k.bz_block(w, b"", used=two, symbols=symbols, **kw))
    stream("bzip2: ran out of selectors", FMT_BZIP2, k.bz_stream([sym([2] * 120 + [3], nsel_field=2)]), 69)
    stream("bzip2: twenty run digits in a row", FMT_BZIP2, k.bz_stream([sym([0] * 20 + [3])]), 70)
    stream("bzip2: nineteen run digits expand beyond the block", FMT_BZIP2, k.bz_stream([sym([1] * 19 + [3])]), 71)
    stream("bzip2: a run of exactly the block size is fine, one more symbol is not", FMT_BZIP2,
           k.bz_stream([sym(run_symbols(100000) + [2, 3])]), 72)
    stream("bzip2: more symbols than the block holds", FMT_BZIP2, k.bz_stream([sym([2] * 100001 + [3])]), 72)
    stream("bzip2: block CRC", FMT_BZIP2, k.bz_stream([blk(crc=0xDEADBEEF)]), 73, detail=0xDEADBEEF)
    stream("bzip2: stream CRC", FMT_BZIP2, k.bz_stream([blk()], stream_crc=0x0BADF00D), 74, detail=0x0BADF00D)
    stream("bzip2: origPtr beyond the block (documented divergence, DESIGN.md)", FMT_BZIP2,
           k.bz_stream([blk(orig=0xFFFFF)]), 75, detail=0xFFFFF)
    good = k.bz_stream([blk()])
    stream("bzip2: footer cut", FMT_BZIP2, good[:-3], 1)
    stream("bzip2: cut inside the symbols", FMT_BZIP2, good[:len(good) // 2], 1)
    stream("bzip2: header only", FMT_BZIP2, b"BZh9", 1)
    stream("bzip2: garbage after the stream", FMT_BZIP2, good + b"junk", 60, detail=0x6A75)
    stream("bzip2: second stream with another level", FMT_BZIP2, good + k.bz_stream([blk()], level=9), 0, out=data * 2)
    stream("bzip2: stream without blocks", FMT_BZIP2, k.bz_stream([]), 0, out=b"")


def run_symbols(run):
    """RUNA / RUNB digits of a zero run (bijective base 2, least significant first)."""
    out = []
    while run:
        run -= 1
        out.append(run & 1)
        run >>= 1
    return out


# =============================================================== LZMA / LZMA2 (README.org:231-308, lzma.lisp, lzma2.lisp)

PROPS = k.lzma_props(3, 0, 2)


def lzma_payload(build, top=False, **kw):
    e = k.LzmaEncoder(**kw)
    build(e)
    return e.finish(top), bytes(e.out)


def lzma_vectors():
    hello = b"hello, hello, hello!"
    hello_syms = lambda e: e.literals(b"hello, ").match(12, 6).literal(ord("!"))
    pay, out = lzma_payload(hello_syms)
    assert out == hello
    chunk = lambda p=pay, n=len(hello), **kw: k.lzma2_lzma_chunk(p, n, props=PROPS, **kw)
    l2 = lambda body: stream_args(body)
    raw2 = lambda name, body, status, **kw: stream(name, FMT_RAW_LZMA2, body, status, opt=1 << 16, **kw)
    raw2("lzma2: one LZMA chunk", chunk() + b"\0", 0, out=hello)
    # ---- range decoder (lzma.lisp:55-73, 174-192)
    raw2("lzma: first byte of the range coder data not zero", chunk(b"\1" + pay[1:]) + b"\0", 80)
    raw2("lzma: initial code 0xFFFFFFFF (liblzma accepts)", chunk(b"\0\xff\xff\xff\xff" + pay[5:]) + b"\0", 81)
    # A direct bit decoded from the last value of an odd range leaves code == bound (lzma.lisp:186-188).  Fresh
    # probabilities (1024) keep the range even, so literals are sent until adapted ones have made it odd at the
    # point where the direct bits of a large distance begin; the bytes written then put `code` at range - 1.
    p82 = direct_bit_trap()
    raw2("lzma: code equals the halved range in a direct bit", k.lzma2_lzma_chunk(p82, 4096, props=PROPS) + b"\0", 82)
    # ---- matches (lzma.lisp:600-612)
    far, _ = lzma_payload(lambda e: e.literals(b"abcdef").match(3, 9))
    raw2("lzma: match distance beyond the data so far", k.lzma2_lzma_chunk(far, 9, props=PROPS) + b"\0", 84)
    wide, wide_out = lzma_payload(lambda e: e.literals(bytes(range(200))).match(4, 149))
    stream("lzma: match distance beyond the dictionary size", FMT_RAW_LZMA2,
           k.lzma2_lzma_chunk(wide, 204, props=PROPS) + b"\0", 83, opt=128)
    stream("lzma: the same chunk with a dictionary of 150", FMT_RAW_LZMA2,
           k.lzma2_lzma_chunk(wide, 204, props=PROPS) + b"\0", 0, out=wide_out, opt=150)
    raw2("lzma: match longer than the chunk's declared size", k.lzma2_lzma_chunk(pay, 15, props=PROPS) + b"\0", 85)
    lit2, _ = lzma_payload(lambda e: e.literals(b"abcd"))
    raw2("lzma2: chunk holds more literals than it declares", k.lzma2_lzma_chunk(lit2, 3, props=PROPS) + b"\0", 89)
    # ---- end of chunk (eof-mode :never, lzma.lisp:640-659, 727-732)
    mk, _ = lzma_payload(lambda e: hello_syms(e).marker())
    raw2("lzma2: end marker at the end of a chunk is illegal", k.lzma2_lzma_chunk(mk, len(hello), props=PROPS) + b"\0", 88)
    mk2, _ = lzma_payload(lambda e: e.literals(b"hel").marker())
    raw2("lzma2: end marker before the declared size", k.lzma2_lzma_chunk(mk2, len(hello), props=PROPS) + b"\0", 88)
    more, _ = lzma_payload(lambda e: hello_syms(e).literals(b"and more"))
    raw2("lzma2: data continues beyond the declared size", k.lzma2_lzma_chunk(more, len(hello), props=PROPS) + b"\0", 89)
    # ---- LZMA2 control bytes (lzma2.lisp:48-204, README.org:259-308)
    raw2("lzma2: empty stream (control 0)", b"\0", 0, out=b"")
    raw2("lzma2: control byte 3", b"\3\0\0", 105, detail=3)
    raw2("lzma2: control byte 0x7F", b"\x7f\0\0", 105, detail=0x7F)
    raw2("lzma2: first chunk without a dictionary reset (LZMA)", chunk(control=0xC0) + b"\0", 102)
    raw2("lzma2: first chunk without a dictionary reset (stored)", k.lzma2_stored_chunk(b"abc", reset_dict=False) + b"\0", 102)
    raw2("lzma2: LZMA chunk after a stored reset needs properties", k.lzma2_stored_chunk(b"abc") +
         k.lzma2_lzma_chunk(pay, len(hello), control=0xA0) + b"\0", 103)
    raw2("lzma2: stored chunks", k.lzma2_stored_chunk(b"abc") + k.lzma2_stored_chunk(b"def", reset_dict=False) + b"\0", 0,
         out=b"abcdef")
    raw2("lzma2: stored (reset), stored (no reset), LZMA chunk that reuses a state that never existed: the code of "
         "lzma2.lisp:157-166 forgets the pending property reset (its own table says P stays P) and then signals a "
         "non-decompression error at lzma2.lisp:107", k.lzma2_stored_chunk(b"abc") +
         k.lzma2_stored_chunk(b"def", reset_dict=False) + k.lzma2_lzma_chunk(pay, len(hello), control=0x80) + b"\0", 107)
    raw2("lzma2: state reset without new properties before any LZMA chunk", k.lzma2_stored_chunk(b"abc") +
         k.lzma2_lzma_chunk(pay, len(hello), control=0x80) + b"\0", 103)
    raw2("lzma2: lc + lp = 5", k.lzma2_lzma_chunk(pay, len(hello), props=k.lzma_props(4, 1, 0)) + b"\0", 100, detail=5)
    raw2("lzma2: pb = 5", k.lzma2_lzma_chunk(pay, len(hello), props=5 * 45) + b"\0", 93, detail=5)
    raw2("lzma2: compressed size larger than the chunk's data, which decodes before the input ends", chunk(csize=len(pay) + 4), 106)
    raw2("lzma2: input ends inside the chunk", chunk()[:-4], 1)
    raw2("lzma2: compressed size smaller than the data", chunk(csize=len(pay) - 2) + b"\0", 104)
    raw2("lzma2: compressed size larger than what the chunk uses", chunk(pay + b"\0\0", csize=len(pay) + 2) + b"\0", 106)
    raw2("lzma2: missing end byte", chunk(), 1)
    raw2("lzma2: stored chunk cut short", k.lzma2_stored_chunk(b"abcdef")[:-2], 1)
    # properties change between chunks; state carried over a stored chunk (README.org:302-308)
    e = k.LzmaEncoder()
    e.literals(b"abcabc")
    p1 = e.finish()
    e.new_chunk()
    e.out += b"XYZ"
    e.total += 3                                       # lzma2.lisp:184-188: stored chunks advance total-i
    e.literal(ord("!")).match(3, 2)
    p2 = e.finish()
    raw2("lzma2: LZMA, stored, LZMA without state reset", k.lzma2_lzma_chunk(p1, 6, props=PROPS) +
         k.lzma2_stored_chunk(b"XYZ", reset_dict=False) + k.lzma2_lzma_chunk(p2, 4, control=0x80) + b"\0", 0, out=bytes(e.out))
    e = k.LzmaEncoder()
    e.literals(b"abcabc")
    p1 = e.finish()
    e.new_chunk()
    e.reset_state(1, 2, 1)
    e.literals(b"defdef").match(5, 2)
    p2 = e.finish()
    raw2("lzma2: second chunk with new properties (lc 1, lp 2, pb 1)", k.lzma2_lzma_chunk(p1, 6, props=PROPS) +
         k.lzma2_lzma_chunk(p2, 11, control=0xC0, props=k.lzma_props(1, 2, 1)) + b"\0", 0, out=bytes(e.out))
    # ---- :lzma2 (dictionary size byte) and .lzma (lzma.lisp:837-865)
    stream("lzma2: dictionary size byte 41", FMT_LZMA2, b"\x29" + chunk() + b"\0", 101, detail=41)
    stream("lzma2: dictionary size byte 40 (4 GiB - 1)", FMT_LZMA2, b"\x28" + chunk() + b"\0", 0, out=hello)
    stream("lzma2: trailing data", FMT_LZMA2, b"\x10" + chunk() + b"\0\0", 170)
    stream(".lzma: with end marker", FMT_LZMA, k.lzma_alone_header() + mk, 0, out=hello)
    mk_top, _ = lzma_payload(lambda e: hello_syms(e).marker(), top=True)
    stream(".lzma: marker found, but the range coder does not end at zero (lzma.lisp:772-774)", FMT_LZMA,
           k.lzma_alone_header() + mk_top, 91)
    stream(".lzma: pb = 5 in the properties byte", FMT_LZMA, k.lzma_alone_header(props=5 * 45) + mk, 93, detail=5)
    stream(".lzma: header cut", FMT_LZMA, k.lzma_alone_header()[:7], 1)
    stream(".lzma: data ends without a marker (the declared size is ignored as written, DESIGN.md section 7)", FMT_LZMA,
           k.lzma_alone_header(size=len(hello)) + pay, 1)
    # ---- :raw-lzma eof modes through the unit route (lzma.lisp:575-577, 640-659)
    lz = dict(lc=3, lp=0, pb=2, dict_size=1 << 16)
    unit("raw-lzma :always with a marker", 6, mk, 0, out=hello, eof_mode=1, cap=64, **lz)
    unit("raw-lzma :always, size reached, no marker follows", 6, pay + bytes(4), 87, eof_mode=1, size=len(hello), cap=64, **lz)
    unit("raw-lzma :never with a marker at the size", 6, mk, 88, eof_mode=2, size=len(hello), cap=64, **lz)
    unit("raw-lzma :never, data continues", 6, more, 89, eof_mode=2, size=len(hello), cap=64, **lz)
    unit("raw-lzma :maybe, size reached, code not zero, no marker", 6, more, 90, eof_mode=0, size=len(hello), cap=64, **lz)
    unit("raw-lzma :maybe, marker before the declared size", 6, mk2, 92, eof_mode=0, size=len(hello), cap=64, **lz)
    unit("raw-lzma: literal beyond the output span", 6, mk, 200, eof_mode=0, cap=5, **lz)


def direct_bit_trap():
    """Range coder bytes that stop in front of the direct bits of a distance with an odd `range` and `code` =
    range - 1.  The range is even whenever a fresh probability (1024) was used last, so short matches and
    literals adapt the length / distance-slot models first; the sequence is searched once, in a fixed order."""
    import random
    for seed in range(4000):
        rng = random.Random(seed)
        e = k.LzmaEncoder()
        e.literals(bytes(rng.randrange(4) for _ in range(8)))
        for _ in range(rng.randrange(20, 120)):
            if rng.random() < 0.5:
                e.match(2, rng.randrange(0, 4))
            else:
                e.literal(rng.randrange(4))
        e.match(2, 0xC0000010, stop_before_direct=True)      # distance slot 63: every slot bit is a one
        if e.rc.range & 1:
            return e.finish(top=True) + b"\xff" * 8          # (0xFF keeps code at range - 1 across normalisations)
    raise AssertionError("no odd range found")


def stream_args(body):
    return body


# =============================================================== XZ (README.org:312-354, xz.lisp)

def xz_vectors():
    plain = b"hello, hello, hello!"
    pay, _ = lzma_payload(lambda e: e.literals(b"hello, ").match(12, 6).literal(ord("!")))
    l2 = k.lzma2_lzma_chunk(pay, len(plain), props=PROPS) + b"\0"
    xz = lambda name, data, status, **kw: stream(name, FMT_XZ, data, status, **kw)
    good = k.xz_file(plain, l2)
    assert lzma.decompress(good) == plain
    xz("xz: valid one-block stream, CRC32", good, 0, out=plain)
    xz("xz: check type none", k.xz_file(plain, l2, check=0), 0, out=plain)
    # stream header (xz.lisp:78-112)
    xz("xz: magic", k.xz_file(plain, l2, header=k.xz_stream_header(magic=b"\xfd7zXZ\x01")), 120)
    xz("xz: reserved header byte", k.xz_file(plain, l2, header=k.xz_stream_header(flags=b"\1\1")), 121)
    xz("xz: header CRC", k.xz_file(plain, l2, header=k.xz_stream_header(crc=b"\0\0\0\0")), 122)
    xz("xz: reserved flag bits", k.xz_file(plain, l2, header=k.xz_stream_header(flags=b"\0\x11")), 123)
    xz("xz: check type 2 (reserved)", k.xz_file(plain, l2, header=k.xz_stream_header(flags=b"\0\2")), 124, detail=2)
    xz("xz: header cut", good[:9], 1)
    # block header (xz.lisp:260-379, 389-407)
    bh = lambda **kw: k.xz_block_header(kw.pop("filters", [(0x21, bytes([16]))]), **kw)
    xz("xz: block header CRC", k.xz_file(plain, l2, block_header=bh(crc=b"\1\2\3\4")), 150, detail=0x04030201)
    xz("xz: reserved block flag bits", k.xz_file(plain, l2, block_header=bh(reserved=0x04)), 137)
    xz("xz: block header padding not zero", k.xz_file(plain, l2, block_header=bh(pad=b"\0\0\1")), 139)
    short = bytes([2, 0, 0x21, 9, 0, 0, 0, 0])              # eight bytes, but the filter says it has nine property bytes
    xz("xz: block header too short for its fields", k.xz_file(plain, l2, block_header=short + k.crc32le(short)), 138)
    xz("xz: non-minimal varint (trailing zero byte)", k.xz_file(plain, l2, block_header=bh(csize=b"\x80\x00")), 127)
    xz("xz: varint longer than nine bytes", k.xz_file(plain, l2, block_header=bh(csize=b"\xff" * 9 + b"\x01")), 128)
    xz("xz: unknown filter id", k.xz_file(plain, l2, block_header=bh(filters=[(0x22, b"")])), 149, detail=0x22)
    xz("xz: delta as the last filter", k.xz_file(plain, l2, block_header=bh(filters=[(0x03, b"\0")])), 141)
    xz("xz: delta with two property bytes", k.xz_file(plain, l2, block_header=bh(filters=[(0x03, b"\0\0"), (0x21, bytes([16]))])), 142, detail=2)
    xz("xz: BCJ as the last filter", k.xz_file(plain, l2, block_header=bh(filters=[(0x04, b"")])), 143)
    xz("xz: BCJ with three property bytes", k.xz_file(plain, l2, block_header=bh(filters=[(0x04, b"\0\0\0"), (0x21, bytes([16]))])), 144, detail=3)
    xz("xz: ARM BCJ start offset not a multiple of four", k.xz_file(plain, l2, block_header=bh(filters=[(0x07, b"\2\0\0\0"), (0x21, bytes([16]))])), 145, detail=2)
    xz("xz: LZMA2 not last", k.xz_file(plain, l2, block_header=bh(filters=[(0x21, bytes([16])), (0x21, bytes([16]))])), 146)
    xz("xz: LZMA2 with two property bytes", k.xz_file(plain, l2, block_header=bh(filters=[(0x21, bytes([16, 0]))])), 147, detail=2)
    xz("xz: LZMA2 reserved property bits", k.xz_file(plain, l2, block_header=bh(filters=[(0x21, bytes([0x50]))])), 148)
    xz("xz: LZMA2 dictionary size byte 41", k.xz_file(plain, l2, block_header=bh(filters=[(0x21, bytes([41]))])), 101, detail=41)
    # delta and x86 chains decode
    delta_in = bytes((plain[i] - (plain[i - 2] if i >= 2 else 0)) & 0xFF for i in range(len(plain)))
    dpay, _ = lzma_payload(lambda e: e.literals(delta_in))
    dl2 = k.lzma2_lzma_chunk(dpay, len(plain), props=PROPS) + b"\0"
    xz("xz: delta (distance 2) + LZMA2", k.xz_file(plain, dl2, filters=[(0x03, b"\1"), (0x21, bytes([16]))]), 0, out=plain)
    # block body (xz.lisp:303-309, 426-457)
    xz("xz: declared compressed size too small", k.xz_file(plain, l2, block_header=bh(csize=len(l2) - 3)), 140)
    xz("xz: declared compressed size too large", k.xz_file(plain, l2, block_header=bh(csize=len(l2) + 1)), 152)
    xz("xz: declared uncompressed size wrong", k.xz_file(plain, l2, block_header=bh(usize=len(plain) + 1)), 153)
    xz("xz: declared sizes right", k.xz_file(plain, l2, block_header=bh(csize=len(l2), usize=len(plain))), 0, out=plain)
    xz("xz: block padding not zero", k.xz_file(plain, l2, block_pad=b"\0" * ((-len(l2) % 4) - 1) + b"\1" if len(l2) % 4 else b""),
       154 if len(l2) % 4 else 0)
    xz("xz: block check wrong", k.xz_file(plain, l2, check_bytes=b"\1\2\3\4"), 155, detail=0x04030201)
    # index and footer (xz.lisp:166-233)
    unp = 12 + len(l2) + 4
    xz("xz: index lists one block too many", k.xz_file(plain, l2, index=k.xz_index([(unp, len(plain))] * 2)), 130)
    xz("xz: index has the wrong uncompressed size", k.xz_file(plain, l2, index=k.xz_index([(unp, len(plain) + 1)])), 131)
    xz("xz: index has the wrong unpadded size", k.xz_file(plain, l2, index=k.xz_index([(unp + 1, len(plain))])), 131)
    plain200 = bytes(range(200))
    p200, _ = lzma_payload(lambda e: e.literals(plain200))
    l200 = k.lzma2_lzma_chunk(p200, 200, props=PROPS) + b"\0"
    unp200 = 12 + len(l200) + 4
    xz("xz: index padding not zero", k.xz_file(plain200, l200, index=k.xz_index([(unp200, 200)], pad=bad_pad(unp200, 200))), 129)
    xz("xz: index CRC", k.xz_file(plain, l2, index=k.xz_index([(unp, len(plain))], crc=b"\0\0\0\0")), 132, detail=0)
    idx = k.xz_index([(unp, len(plain))])
    xz("xz: footer CRC", k.xz_file(plain, l2, footer=k.xz_footer(len(idx), crc=b"\4\3\2\1")), 133, detail=0x01020304)
    xz("xz: backward size", k.xz_file(plain, l2, footer=k.xz_footer(len(idx), backward=7)), 134)
    xz("xz: footer flags differ from the header's", k.xz_file(plain, l2, footer=k.xz_footer(len(idx), check=4)), 135)
    xz("xz: footer magic", k.xz_file(plain, l2, footer=k.xz_footer(len(idx), magic=b"YY")), 136)
    xz("xz: non-minimal varint in the index", k.xz_file(plain, l2, index=k.xz_index([], count=1, raw_records=b"\x80\x00" + k.xz_varint(len(plain)))), 127)
    xz("xz: footer cut", good[:-5], 1)
    # stream padding and multiple streams (xz.lisp:118-134)
    xz("xz: padding of four and eight zero bytes, second stream", good + bytes(4) + good + bytes(8), 0, out=plain * 2)
    xz("xz: padding of three bytes", good + bytes(3), 126)
    xz("xz: padding of six bytes", good + bytes(6), 126)
    xz("xz: garbage after the stream", good + b"garbage!", 125)
    xz("xz: second stream's magic damaged after four bytes", good + b"\xfd7zXY\x00" + good[6:], 125)
    xz("xz: stream without blocks", k.xz_stream_header(1) + k.xz_index([]) + k.xz_footer(len(k.xz_index([])), 1), 0, out=b"")


def bad_pad(unpadded, usize):
    body = 2 + len(k.xz_varint(unpadded)) + len(k.xz_varint(usize))
    n = -body % 4
    assert n, "this record needs no padding"
    return b"\0" * (n - 1) + b"\1"


deflate_vectors()
bzip2_vectors()
lzma_vectors()
xz_vectors()

# statuses no input can provoke through these routes, and why (the GPU test takes them out of its coverage check)
UNREACHABLE = {
    86: "SEMZ_E_LZMA_LITERAL_SIZE: the literal check of lzma.lisp:717-718 sits behind the p = end test of the loop head",
    151: "SEMZ_E_XZ_TOTAL_SIZE: needs 2^63 bytes of output",
    156: "SEMZ_E_XZ_STREAM_TOO_LONG: needs 2^63 bytes of input",
    202: "SEMZ_E_NOT_RUN: reserved for chained units, no kernel sets it",
}
# engine-level conditions without a counterpart in the reference, covered by their own tests
ENGINE_LEVEL = {
    201: "SEMZ_E_UNSUPPORTED: tests/test_gpu_lzma.py (wide literal contexts), tests/test_gpu_xz.py (two BCJ offsets)",
    203: "SEMZ_E_BAD_UNIT: tests/test_gpu_device_entry.py",
    204: "SEMZ_E_NO_MEMORY: needs a device-memory exhaustion; the size clamps are in tests/test_gpu_untrusted_sizes.py",
}
