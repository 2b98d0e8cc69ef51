"""A small LZMA1 encoder for tests (test infrastructure, nothing here ships).

liblzma only writes lc + lp <= 4 and always ends raw LZMA1 data with a marker, so it cannot
produce two kinds of input the reference accepts (lzma.lisp:809-835): wide literal contexts
(lc up to 8, lp up to 4) and data of a declared size without a marker.  This encoder follows the
published LZMA format (Igor Pavlov's lzma-specification.txt): greedy matches from a 3-byte hash,
rep0 matches and short reps so that every symbol kind of decode-lzma (lzma.lisp:708-765) appears.
"""
from collections import defaultdict


class _RangeEncoder:
    def __init__(self):
        self.low, self.range, self.cache, self.cache_size = 0, 0xFFFFFFFF, 0, 1
        self.out = bytearray()
        self.probs = defaultdict(lambda: 1024)

    def _shift_low(self):
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

    def bit(self, key, b):
        p = self.probs[key]
        bound = (self.range >> 11) * p
        if b == 0:
            self.range = bound
            self.probs[key] = p + ((2048 - p) >> 5)
        else:
            self.low += bound
            self.range -= bound
            self.probs[key] = p - (p >> 5)
        while self.range < 1 << 24:
            self.range = (self.range << 8) & 0xFFFFFFFF
            self._shift_low()

    def direct(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.range >>= 1
            if (value >> i) & 1:
                self.low += self.range
            while self.range < 1 << 24:
                self.range = (self.range << 8) & 0xFFFFFFFF
                self._shift_low()

    def tree(self, key, nbits, value):
        m = 1
        for i in range(nbits - 1, -1, -1):
            b = (value >> i) & 1
            self.bit((key, m), b)
            m = (m << 1) | b

    def rtree(self, key, nbits, value):
        m = 1
        for i in range(nbits):
            b = (value >> i) & 1
            self.bit((key, m), b)
            m = (m << 1) | b

    def finish(self):
        for _ in range(5):
            self._shift_low()
        return bytes(self.out)


def encode(data, lc=3, lp=0, pb=2, dict_size=1 << 16, marker=True, seed=1):
    """Raw LZMA1 data (range coder bytes only, no header)."""
    import random
    rng = random.Random(seed)
    rc = _RangeEncoder()
    n = len(data)
    state, rep0, pos = 0, 0, 0
    have_rep = False
    table = {}

    def length(kind, ln, ps):
        ln -= 2
        if ln < 8:
            rc.bit((kind, "c0"), 0)
            rc.tree((kind, "low", ps), 3, ln)
        elif ln < 16:
            rc.bit((kind, "c0"), 1)
            rc.bit((kind, "c1"), 0)
            rc.tree((kind, "mid", ps), 3, ln - 8)
        else:
            rc.bit((kind, "c0"), 1)
            rc.bit((kind, "c1"), 1)
            rc.tree((kind, "high"), 8, ln - 16)

    def distance(dist, ln):
        ls = min(ln - 2, 3)
        if dist < 4:
            slot = dist
        else:
            hb = dist.bit_length() - 1
            slot = 2 * hb + ((dist >> (hb - 1)) & 1)
        rc.tree(("slot", ls), 6, slot)
        if slot >= 4:
            fb = (slot >> 1) - 1
            base = (2 | (slot & 1)) << fb
            red = dist - base
            if slot < 14:
                rc.rtree(("spec", base - slot), fb, red)
            else:
                rc.direct(red >> 4, fb - 4)
                rc.rtree(("align",), 4, red & 15)

    def match_len(a, b, limit):
        k = 0
        while k < limit and data[a + k] == data[b + k]:
            k += 1
        return k

    while pos < n:
        ps = pos & ((1 << pb) - 1)
        # candidates
        best_len, best_dist = 0, 0
        if pos + 3 <= n:
            key = bytes(data[pos:pos + 3])
            cand = table.get(key)
            if cand is not None and pos - cand <= dict_size:
                best_len = match_len(cand, pos, min(273, n - pos))
                best_dist = pos - cand - 1
            table[key] = pos
        rep_len = match_len(pos - rep0 - 1, pos, min(273, n - pos)) if have_rep and pos - rep0 - 1 >= 0 else 0
        if rep_len >= 2 and rep_len + 1 >= best_len:
            rc.bit(("islit", state, ps), 1)
            rc.bit(("issimple", state), 1)
            rc.bit(("rep0", state), 0)
            rc.bit(("short", state, ps), 1)
            length("replen", rep_len, ps)
            state = 8 if state < 7 else 11
            pos += rep_len
        elif rep_len == 1 and best_len < 2 and rng.random() < 0.5:
            rc.bit(("islit", state, ps), 1)
            rc.bit(("issimple", state), 1)
            rc.bit(("rep0", state), 0)
            rc.bit(("short", state, ps), 0)
            state = 9 if state < 7 else 11
            pos += 1
        elif best_len >= 2 and (best_len >= 3 or best_dist < 128):
            rc.bit(("islit", state, ps), 1)
            rc.bit(("issimple", state), 0)
            length("len", best_len, ps)
            distance(best_dist, best_len)
            rep0, have_rep = best_dist, True
            state = 7 if state < 7 else 10
            pos += best_len
        else:
            rc.bit(("islit", state, ps), 0)
            prev = data[pos - 1] if pos else 0
            ctx = ((pos & ((1 << lp) - 1)) << lc) | (prev >> (8 - lc))
            byte, sym = data[pos], 1
            if state >= 7:
                mb = data[pos - rep0 - 1]
                for i in range(7, -1, -1):
                    m, b = (mb >> i) & 1, (byte >> i) & 1
                    rc.bit(("lit", ctx, 0x100 + (m << 8) + sym), b)
                    sym = (sym << 1) | b
                    if m != b:
                        break
            while sym < 0x100:
                b = (byte >> (7 - (sym.bit_length() - 1))) & 1
                rc.bit(("lit", ctx, sym), b)
                sym = (sym << 1) | b
            state = 0 if state < 4 else state - 3 if state < 10 else state - 6
            pos += 1
    if marker:
        ps = pos & ((1 << pb) - 1)
        rc.bit(("islit", state, ps), 1)
        rc.bit(("issimple", state), 0)
        length("len", 2, ps)
        distance(0xFFFFFFFF, 2)
    return rc.finish()


def alone(data, lc=3, lp=0, pb=2, dict_size=1 << 16, marker=True, declare_size=False, seed=1):
    """A .lzma file: properties byte, dictionary size, size (or all ones), raw LZMA1 data."""
    head = bytes([(pb * 5 + lp) * 9 + lc]) + dict_size.to_bytes(4, "little")
    head += (len(data) if declare_size else (1 << 64) - 1).to_bytes(8, "little")
    return head + encode(data, lc, lp, pb, dict_size, marker, seed)
