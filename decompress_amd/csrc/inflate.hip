// inflate.hip -- gfx950 kernel for Deflate / zlib / gzip units.
//
// Replaces, per unit, the reference loops
//   read-dynamic-dhi            deflate.lisp:160-201
//   lengths->ht                 huffman.lisp:142-225   (canonical tables, exact-completeness check)
//   decode-huffman-data         deflate.lisp:272-338
//   decode-uncompressed-data    deflate.lisp:341-355
//   next-deflate-chunk          deflate.lisp:365-410   (block state machine)
//   zlib / gzip header+trailer  zlib.lisp:21-78, gzip.lisp:31-115
//
// Layout: one wave per workgroup, 64/G units per wave (semz_device.h).  The
// group's leader decodes Huffman symbols from a 64-bit bit buffer into a
// small token queue in LDS; then all G lanes resolve the tokens: literals
// and matches whose source predates the round are written lane-parallel,
// matches that depend on bytes of the same round are replayed in order with
// a group-wide copy (periodic replication when distance < length,
// common.lisp:324-333).  Output is staged in an LDS ring and flushed to HBM
// with 16-byte stores; sources older than the ring are read back from HBM.
// Huffman tables are rebuilt in LDS per block: a root lookup table plus a
// canonical (first-code / sorted-symbol) description used for codes longer
// than the root and for the table fill itself.
#include "semz_device.h"

namespace {

constexpr int LIT_ROOT_BITS = 10;
constexpr int DIST_ROOT_BITS = 8;
constexpr int CLEN_ROOT_BITS = 7;
constexpr int RING = 4096;               // output staging ring per unit (bytes)
constexpr int RING_MASK = RING - 1;
constexpr int IN_RING = 2048;            // compressed-input staging ring per unit (bytes)
constexpr int ROUND_BYTES = 2048;        // a round stops once it has produced this many bytes
constexpr int MIN_BLOCK_BITS = 10;       // +min-deflate-block-bitsize+, deflate.lisp:4-12

enum { T_LIT = 0, T_MATCH = 1, T_RAW = 2 };
enum { EV_ROUND = 0, EV_DYN = 1, EV_FIXED = 2, EV_DONE = 3 };
enum { ST_HDR = 0, ST_HUFF = 1, ST_STORED = 2 };
enum { MODE_NORMAL = 0, MODE_LITONLY = 1, MODE_SINGLE = 2 };

// lit/len root entry:  [3:0] code length (0 = longer than root) [5:4] kind
//   kind 0 literal: [15:8] byte
//   kind 1 length : [10:8] extra bits, [19:11] base length, [24:20] symbol-257
//   kind 2 end of block
//   kind 3 invalid symbol 286/287: [16:8] symbol
// dist root entry:     [3:0] code length, [4] invalid (30/31), [9:5] code,
//                      [13:10] extra bits, [31:16] base distance
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_BAD = 3 };

struct TreeMeta {
    uint16_t count[16];
    uint16_t first[16];   // first canonical code (MSB-first) per length
    uint16_t offs[16];    // start of each length's run in sorted[]
    uint32_t limit[16];   // (first+count) << (15-len): exclusive bound of the length, left aligned to 15 bits
    uint32_t minlen, maxlen, status, pad;
};

template <int G> struct __align__(16) UnitLds {
    uint32_t lit_root[1 << LIT_ROOT_BITS];
    uint32_t dist_root[1 << DIST_ROOT_BITS];
    uint32_t queue[2 * G];
    uint8_t ring[RING];
    uint8_t inring[IN_RING];
    uint16_t lit_sorted[288];
    uint8_t dist_sorted[32];
    uint8_t clen_sorted[32];
    uint8_t clen_root[1 << CLEN_ROOT_BITS];
    uint8_t lens[320];
    uint8_t clens[32];
    TreeMeta lit, dist, clen;
};

__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t len)
{
    if (sym < 256) return (sym << 8) | (K_LIT << 4) | len;
    if (sym == 256) return (K_EOB << 4) | len;
    if (sym > 285) return (sym << 8) | (K_BAD << 4) | len;
    uint32_t i = sym - 257, extra, base;                 // deflate.lisp:14-23
    if (i < 8) { extra = 0; base = 3 + i; }
    else if (i == 28) { extra = 0; base = 258; }
    else { extra = (i >> 2) - 1; base = 3 + ((4 + (i & 3)) << extra); }
    return (i << 20) | (base << 11) | (extra << 8) | (K_LEN << 4) | len;
}

__device__ __forceinline__ uint32_t dist_entry(uint32_t code, uint32_t len)
{
    if (code > 29) return (code << 5) | (1u << 4) | len;
    uint32_t extra, base;                                // deflate.lisp:25-34
    if (code < 4) { extra = 0; base = code + 1; }
    else { extra = (code >> 1) - 1; base = 1 + ((2 + (code & 1)) << extra); }
    return (base << 16) | (extra << 10) | (code << 5) | len;
}

__device__ __forceinline__ uint32_t clen_entry(uint32_t sym, uint32_t len) { return (sym << 3) | len; }

// Canonical decode of a left-aligned 15-bit MSB-first window, lengths lo..15.
// Returns the code length (0 if none in range) and the symbol's index in sorted[].
__device__ __forceinline__ uint32_t canon_find(const TreeMeta &m, uint32_t v15, int lo, int hi, uint32_t &idx)
{
    for (int l = lo; l <= hi; l++) {
        if (v15 < m.limit[l]) {
            idx = m.offs[l] + (v15 >> (15 - l)) - m.first[l];
            return (uint32_t)l;
        }
    }
    return 0;
}

// Group-parallel construction of one canonical tree (lengths->ht,
// huffman.lisp:142-225): per-length counts (one lane per length), the
// exact-completeness check (huffman.lisp:166-184), the sorted symbol list and
// the root table.  sorted[] entries are SortT; root entries RootT.
template <int G, int ROOT_BITS, typename SortT, typename RootT, typename MakeEntry>
__device__ void build_tree(const uint8_t *lens, int n, TreeMeta &m, SortT *sorted, RootT *root,
                           MakeEntry make_entry)
{
    using Gr = Group<G>;
    const int gl = Gr::gl();
    static_assert(G >= 16, "one lane per code length");
    // 1. counts
    uint32_t c = 0;
    if (gl >= 1 && gl < 16)
        for (int s = 0; s < n; s++) c += (lens[s] == gl);
    if (gl < 16) m.count[gl] = (uint16_t)c;
    lds_sync();
    // 2. canonical bookkeeping by the leader
    if (gl == 0) {
        uint32_t code = 0, off = 0, total = 0, minl = 20, maxl = 1;
        m.first[0] = 0; m.offs[0] = 0; m.limit[0] = 0;
        for (int l = 1; l <= 15; l++) {
            uint32_t cnt = m.count[l];
            code <<= 1;
            m.first[l] = (uint16_t)code;
            m.offs[l] = (uint16_t)off;
            code += cnt;
            off += cnt;
            m.limit[l] = code << (15 - l);
            total += cnt << (15 - l);
            if (cnt) { if ((uint32_t)l < minl) minl = l; maxl = l > 1 ? l : 1; }
        }
        m.minlen = minl;
        m.maxlen = maxl;
        m.status = total < 32768u ? SEMZ_E_HUFF_UNDERFULL : total > 32768u ? SEMZ_E_HUFF_OVERFULL : SEMZ_OK;
    }
    lds_sync();
    if (m.status != SEMZ_OK) return;
    // 3. sorted symbol list: lane L appends its symbols in ascending order
    if (gl >= 1 && gl < 16) {
        uint32_t k = m.offs[gl];
        for (int s = 0; s < n; s++)
            if (lens[s] == gl) sorted[k++] = (SortT)s;
    }
    lds_sync();
    // 4. root table, one index per lane per step
    for (int i = gl; i < (1 << ROOT_BITS); i += G) {
        uint32_t v15 = (__brev((uint32_t)i) >> (32 - ROOT_BITS)) << (15 - ROOT_BITS);
        uint32_t idx = 0;
        uint32_t l = canon_find(m, v15, 1, ROOT_BITS, idx);
        root[i] = l ? (RootT)make_entry((uint32_t)sorted[idx], l) : (RootT)0;
    }
    lds_sync();
}

// LSB-first bit reader (lsb-bit-reader, bits.lisp:118) over the unit's LDS
// input ring.  Positions are "v-offsets": byte offset inside `in` plus the
// 16-byte misalignment of the `in` pointer, so that v % 16 == address % 16 and
// the ring can be filled with aligned 16-byte loads (stage_input).  The ring
// holds zero bytes past the end of the unit's input; near_end turns on the
// exact end-of-input accounting that decides `eof` (bits.lisp:35-45).
struct BitReader {
    const uint8_t *ring;
    uint64_t wpos;    // v-offset of the next dword to take from the ring (multiple of 4)
    uint64_t limit;   // v-offset of the first byte that does not exist
    uint64_t buf;
    int cnt;          // bits in buf
    int skip;         // bits to discard from the first dword (unaligned start)
    bool near_end;

    __device__ __forceinline__ void init(const uint8_t *r, uint64_t v, uint64_t lim)
    {
        ring = r; wpos = v & ~3ull; limit = lim; buf = 0; cnt = 0; skip = (int)(v & 3) * 8; near_end = false;
    }
    __device__ __forceinline__ uint32_t next_dword()
    {
        uint32_t w = *reinterpret_cast<const uint32_t *>(ring + (wpos & (IN_RING - 1)));
        if (wpos + 4 > limit) near_end = true;
        wpos += 4;
        return w;
    }
    // after refill(): cnt >= 33
    __device__ __forceinline__ void refill()
    {
        if (skip) {
            buf = (uint64_t)next_dword() >> skip;
            cnt = 32 - skip;
            skip = 0;
        }
        if (cnt <= 32) {
            buf |= (uint64_t)next_dword() << cnt;
            cnt += 32;
        }
    }
    // bits left between the read position and the end of the input
    __device__ __forceinline__ int64_t remaining() const
    {
        return (int64_t)(8 * limit) - (int64_t)(8 * wpos) + cnt - skip;
    }
    __device__ __forceinline__ bool short_of(int n) const { return near_end && remaining() < n; }
    // bit position as a v-offset in bits
    __device__ __forceinline__ uint64_t bitpos() const { return 8 * wpos - (uint64_t)cnt + (uint64_t)skip; }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ void align_byte() { int d = cnt & 7; buf >>= d; cnt -= d; }
};

// Group-cooperative top-up of the input ring with aligned 16-byte loads:
// after the call at least IN_RING - 16*G staged bytes lie ahead of wpos.
// Bytes outside [in_lo, limit) (v-offsets) are staged as zero.
template <int G>
__device__ __forceinline__ void stage_input(uint8_t *ring, const uint8_t *in_v0, uint64_t in_lo, uint64_t limit,
                                            uint64_t wpos, uint64_t &s_hi)
{
    const int gl = Group<G>::gl();
    const uint64_t wfloor = wpos & ~15ull;
    if (s_hi < wfloor || s_hi > wfloor + IN_RING) s_hi = wfloor;      // first call, or the reader jumped ahead
    while (s_hi + 16u * G <= wfloor + IN_RING) {
        const uint64_t v = s_hi + 16u * gl;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (v >= in_lo && v + 16 <= limit) {
            x = *reinterpret_cast<const uint4 *>(in_v0 + v);
        } else if (v < limit && v + 16 > in_lo) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16; b++) {
                uint64_t vv = v + b;
                if (vv >= in_lo && vv < limit) w[b >> 2] |= (uint32_t)in_v0[vv] << (8 * (b & 3));
            }
            x = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(ring + (v & (IN_RING - 1))) = x;
        s_hi += 16u * G;
    }
    lds_sync();
}

struct Status {
    int32_t code;
    uint32_t detail;
    __device__ __forceinline__ void set(int c, uint32_t d) { if (code == SEMZ_OK) { code = c; detail = d; } }
};

// bitwise CRC-32 for the gzip header check (gzip.lisp:73-78); headers are tiny
__device__ uint32_t crc32_bitwise(const uint8_t *p, uint64_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
    }
    return ~c;
}

// gzip header, gzip.lisp:31-87.  Returns the offset of the deflate data.
__device__ uint64_t parse_gzip_header(const uint8_t *in, uint64_t p0, uint64_t lim, Status &st)
{
    uint64_t p = p0;
    if (lim - p < 10) { st.set(SEMZ_EOF, 0); return p; }
    uint32_t flg = in[p + 3];
    if (in[p] != 0x1F || in[p + 1] != 0x8B) { st.set(SEMZ_E_GZIP_MAGIC, 0); return p; }
    if (in[p + 2] != 8) { st.set(SEMZ_E_GZIP_METHOD, in[p + 2]); return p; }
    if (flg >> 5) { st.set(SEMZ_E_GZIP_RESERVED_FLAGS, 0); return p; }
    p += 10;
    if (flg & 4) {
        if (lim - p < 2) { st.set(SEMZ_EOF, 0); return p; }
        uint64_t xl = in[p] | ((uint32_t)in[p + 1] << 8);
        p += 2;
        if (lim - p < xl) { st.set(SEMZ_EOF, 0); return p; }
        uint64_t i = 0;                                  // gzip-extra-fields-consistent-p
        while (i < xl) {
            if (i + 4 > xl) { st.set(SEMZ_E_GZIP_EXTRA_FIELDS, 0); return p; }
            uint64_t l = in[p + i + 2] | ((uint32_t)in[p + i + 3] << 8);
            if (i + 4 + l > xl) { st.set(SEMZ_E_GZIP_EXTRA_FIELDS, 0); return p; }
            i += 4 + l;
        }
        p += xl;
    }
    for (int which = 0; which < 2; which++) {
        if (!(flg & (which ? 16 : 8))) continue;
        for (;;) {
            if (p >= lim) { st.set(SEMZ_EOF, 0); return p; }
            if (in[p++] == 0) break;
        }
    }
    if (flg & 2) {
        if (lim - p < 2) { st.set(SEMZ_EOF, 0); return p; }
        uint32_t want = in[p] | ((uint32_t)in[p + 1] << 8);
        uint32_t got = crc32_bitwise(in + p0, p - p0) & 0xFFFF;
        p += 2;
        if (want != got) { st.set(SEMZ_E_GZIP_HEADER_CRC, want); return p; }
    }
    return p;
}

template <int G>
__global__ __launch_bounds__(64) void inflate_kernel(const uint8_t *in, const semz_unit *units,
                                                     uint32_t nunits, uint8_t *out,
                                                     semz_result *results, semz_tail *tails)
{
    using Gr = Group<G>;
    constexpr int UPW = 64 / G;                      // units per wave
    __shared__ UnitLds<G> lds_all[UPW];
    const int gl = Gr::gl();
    const bool leader = gl == 0;
    const uint32_t u = blockIdx.x * UPW + (Gr::lane() / G);
    const bool have_unit = u < nunits;
    UnitLds<G> &L = lds_all[Gr::lane() / G];

    semz_unit un;
    if (have_unit) un = units[u];
    else { un = semz_unit{}; }

    uint8_t *ob = out + un.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);      // ring index == global address (mod 16)
    const uint64_t out_cap = un.out_cap;

    // ---- leader-only state ----
    // v-offsets (see BitReader): offset inside `in` + misalignment of the `in` pointer
    const uint64_t amis = (uint64_t)((uintptr_t)in & 15);
    const uint8_t *in_v0 = in - amis;                // in_v0 + v is the address of v-offset v
    const uint64_t limit_v = un.in_limit + amis;
    uint64_t s_hi = 0;                               // input staged up to this v-offset (group-uniform)
    BitReader br;
    Status st{SEMZ_OK, 0};
    br.init(L.inring, un.in_off + amis, limit_v);
    uint32_t wsize = un.param0 ? un.param0 : 32768u;
    uint64_t prefix_amount = 0;
    const uint8_t *prefix_end = in;      // one past the last history byte
    int state = ST_HDR;
    bool final_block = false;
    bool finished = false;               // final block fully decoded
    uint32_t stored_left = 0;
    uint32_t aux = 0;
    int dist_mode = MODE_NORMAL;
    bool lit_full = false, dist_full = false;   // ht-full-read-p, huffman.lisp:163
    int lit_max = 15, dist_max = 15;
    int nlit = 0, ndist = 0;
    int trailing_bits = un.kind == SEMZ_UNIT_GZIP ? 64 : un.kind == SEMZ_UNIT_ZLIB ? 32 : 0;

    uint64_t pos = 0;                    // output bytes produced (all lanes track it)
    uint64_t flushed = 0;                // output bytes already in HBM
    bool alive = have_unit;

    // ---- container header (leader) ----
    if (leader && alive) {
        if (un.kind == SEMZ_UNIT_GZIP) {
            uint64_t p = parse_gzip_header(in, un.in_off, un.in_limit, st);
            aux = (uint32_t)(p - un.in_off);
            br.init(L.inring, p + amis, limit_v);
            wsize = 32768u;
        } else if (un.kind == SEMZ_UNIT_ZLIB) {          // zlib.lisp:21-57
            uint64_t p = un.in_off;
            if (un.in_limit - p < 2) st.set(SEMZ_EOF, 0);
            else {
                uint32_t cmf = in[p], flg = in[p + 1];
                p += 2;
                if ((cmf & 15) != 8) st.set(SEMZ_E_ZLIB_METHOD, cmf & 15);
                else if ((cmf >> 4) > 7) st.set(SEMZ_E_ZLIB_CINFO, cmf >> 4);
                else if ((cmf * 256 + flg) % 31) st.set(SEMZ_E_ZLIB_FCHECK, 0);
                else {
                    wsize = 1u << ((cmf >> 4) + 8);
                    if (flg & 32) {
                        if (un.in_limit - p < 4) st.set(SEMZ_EOF, 0);
                        else {
                            aux = ((uint32_t)in[p] << 24) | ((uint32_t)in[p + 1] << 16) |
                                  ((uint32_t)in[p + 2] << 8) | in[p + 3];
                            p += 4;
                            if (!(un.flags & SEMZ_UF_HAVE_DICT_FN)) st.set(SEMZ_E_ZLIB_NO_DICT_FN, 0);
                            else if (un.prefix_len == 0 && un.prefix_off == 0) st.set(SEMZ_E_ZLIB_UNKNOWN_DICT, aux);
                        }
                    }
                }
                br.init(L.inring, p + amis, limit_v);
            }
        }
        if (un.prefix_len) {                              // deflate.lisp:419-425
            prefix_amount = un.prefix_len < wsize ? un.prefix_len : wsize;
            prefix_end = in + un.prefix_off + un.prefix_len;
        }
    }
    prefix_amount = Gr::bcast(prefix_amount);
    {
        uint64_t pe = Gr::bcast((uint64_t)(uintptr_t)prefix_end);
        prefix_end = (const uint8_t *)(uintptr_t)pe;
    }

    // ---- main loop ----
    while (__any(alive)) {
        int ev = EV_DONE;
        uint32_t ntok = 0, rbytes = 0;
        // top up the input ring ahead of the reader (all lanes)
        {
            uint64_t wp = Gr::bcast(br.wpos);
            if (alive) stage_input<G>(L.inring, in_v0, amis, limit_v, wp, s_hi);
        }
        if (leader && alive) {
            ev = EV_ROUND;
            if (st.code != SEMZ_OK) ev = EV_DONE;
            // every iteration takes at most five dwords from the ring
            while (ev == EV_ROUND && ntok < (uint32_t)G && rbytes < (uint32_t)ROUND_BYTES &&
                   br.wpos + 32 <= s_hi) {
                br.refill();
                if (state == ST_HDR) {
                    if (final_block) { finished = true; ev = EV_DONE; break; }
                    // BFINAL, BTYPE: deflate.lisp:374-397
                    if (br.short_of(1)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                    final_block = br.peek(1); br.drop(1);
                    if (br.short_of(2)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                    uint32_t type = br.peek(2); br.drop(2);
                    if (type == 0) {
                        br.align_byte();
                        br.refill();
                        if (br.short_of(16)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        uint32_t len = br.peek(16); br.drop(16);
                        if (br.short_of(16)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        uint32_t chk = br.peek(16); br.drop(16);
                        if (chk != (len ^ 0xFFFFu)) { st.set(SEMZ_E_DEFLATE_STORED_LEN, chk); ev = EV_DONE; break; }
                        stored_left = len;
                        state = ST_STORED;
                        if (len == 0) state = ST_HDR;
                        else {
                            // the whole stored block must exist (decode-uncompressed-data, deflate.lisp:350)
                            if (pos + rbytes + len > out_cap) { st.set(SEMZ_E_OUT_CAP, 0); ev = EV_DONE; break; }
                            if (br.remaining() < (int64_t)8 * len) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        }
                    } else if (type == 1) { ev = EV_FIXED; break; }
                    else if (type == 2) {
                        // read-dynamic-dhi, deflate.lisp:160-175: counts and code-length code lengths
                        if (br.short_of(5)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        nlit = 257 + br.peek(5); br.drop(5);
                        if (br.short_of(5)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        ndist = 1 + br.peek(5); br.drop(5);
                        if (br.short_of(4)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        int nclen = 4 + br.peek(4); br.drop(4);
                        for (int i = 0; i < 19; i++) L.clens[i] = 0;
                        bool eof = false;
                        for (int i = 0; i < nclen; i++) {
                            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                            br.refill();
                            if (br.short_of(3)) { eof = true; break; }
                            L.clens[order[i]] = (uint8_t)br.peek(3); br.drop(3);
                        }
                        if (eof) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                        ev = EV_DYN;
                        break;
                    } else { st.set(SEMZ_E_DEFLATE_BTYPE, 0); ev = EV_DONE; break; }
                    continue;
                }
                if (state == ST_STORED) {
                    // raw bytes: the bit buffer is byte aligned; hand the span to the group
                    uint32_t room = ROUND_BYTES - rbytes;
                    uint32_t n = stored_left < room ? stored_left : room;
                    uint64_t src = br.bitpos() >> 3;              // v-offset of the span
                    L.queue[2 * ntok] = T_RAW | (n << 2) | (rbytes << 16);
                    L.queue[2 * ntok + 1] = (uint32_t)(src - amis - un.in_off);
                    ntok++;
                    rbytes += n;
                    stored_left -= n;
                    br.init(L.inring, src + n, limit_v);          // restart the bit buffer after the span
                    if (stored_left == 0) state = ST_HDR;
                    break;                                        // the ring must be restaged at the new position
                }
                // ---- ST_HUFF: one literal/length symbol, decode-huffman-data deflate.lisp:313-338 ----
                uint32_t e = L.lit_root[br.peek(LIT_ROOT_BITS)];
                uint32_t cl = e & 15;
                if (cl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    cl = canon_find(L.lit, v15, LIT_ROOT_BITS + 1, 15, idx);
                    e = lit_entry(L.lit_sorted[idx], cl);
                }
                if (br.short_of(lit_full ? lit_max : (int)cl)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                br.drop(cl);
                uint32_t kind = (e >> 4) & 3;
                if (kind == K_LIT) {
                    if (pos + rbytes >= out_cap) { st.set(SEMZ_E_OUT_CAP, 0); ev = EV_DONE; break; }
                    L.queue[2 * ntok] = T_LIT | (1u << 2) | (rbytes << 16);
                    L.queue[2 * ntok + 1] = (e >> 8) & 0xFF;
                    ntok++;
                    rbytes++;
                    continue;
                }
                if (kind == K_EOB) { state = ST_HDR; continue; }
                if (kind == K_BAD) { st.set(SEMZ_E_DEFLATE_LITLEN_CODE, e >> 8); ev = EV_DONE; break; }
                // length + distance
                if (dist_mode == MODE_LITONLY) { st.set(SEMZ_E_DEFLATE_LEN_IN_LITONLY, 257 + (e >> 20)); ev = EV_DONE; break; }
                uint32_t lx = (e >> 8) & 7;
                if (br.short_of(lx)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                uint32_t length = ((e >> 11) & 0x1FF) + br.peek(lx);
                br.drop(lx);
                br.refill();
                uint32_t d = L.dist_root[br.peek(DIST_ROOT_BITS)];
                uint32_t dl = d & 15;
                if (dl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    dl = canon_find(L.dist, v15, DIST_ROOT_BITS + 1, 15, idx);
                    d = dist_entry(L.dist_sorted[idx], dl);
                }
                if (br.short_of(dist_full ? dist_max : (int)dl)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                br.drop(dl);
                if (d & 16) {
                    st.set(dist_mode == MODE_SINGLE ? SEMZ_E_DEFLATE_SINGLE_DIST : SEMZ_E_DEFLATE_DIST_CODE, (d >> 5) & 31);
                    ev = EV_DONE; break;
                }
                uint32_t dx = (d >> 10) & 15;
                if (br.short_of(dx)) { st.set(SEMZ_EOF, 0); ev = EV_DONE; break; }
                uint32_t distance = (d >> 16) + br.peek(dx);
                br.drop(dx);
                uint64_t pointer = prefix_amount + pos + rbytes;            // deflate.lisp:329-331
                uint64_t lim = pointer < wsize ? pointer : wsize;
                if (distance > lim) { st.set(SEMZ_E_DEFLATE_DISTANCE, distance); ev = EV_DONE; break; }
                if (pos + rbytes + length > out_cap) { st.set(SEMZ_E_OUT_CAP, 0); ev = EV_DONE; break; }
                L.queue[2 * ntok] = T_MATCH | (length << 2) | (rbytes << 16);
                L.queue[2 * ntok + 1] = distance;
                ntok++;
                rbytes += length;
            }
        }
        ev = Gr::bcast(ev);
        ntok = Gr::bcast(ntok);
        rbytes = Gr::bcast(rbytes);
        lds_sync();

        // ---- resolve the round's tokens ----
        const uint64_t pos0 = pos, pos1 = pos + rbytes;
        const int64_t ring_lo = (int64_t)pos1 - RING;      // positions >= ring_lo live in the ring
        if (alive && ntok) {
            bool dep = false;
            if ((uint32_t)gl < ntok) {
                uint32_t w0 = L.queue[2 * gl], w1 = L.queue[2 * gl + 1];
                uint32_t type = w0 & 3, len = (w0 >> 2) & 0x3FFF;
                uint64_t p = pos0 + (w0 >> 16);
                if (type == T_LIT) {
                    L.ring[(uint32_t)(p + shift) & RING_MASK] = (uint8_t)w1;
                } else if (type == T_MATCH) {
                    int64_t src = (int64_t)p - (int64_t)w1, src_end = src + len;
                    if (src >= ring_lo && src >= 0 && src_end <= (int64_t)pos0) {
                        for (uint32_t k = 0; k < len; k++)
                            L.ring[(uint32_t)(p + k + shift) & RING_MASK] = L.ring[(uint32_t)(src + k + shift) & RING_MASK];
                    } else if (src >= 0 && src_end <= (int64_t)flushed) {
                        for (uint32_t k = 0; k < len; k++)
                            L.ring[(uint32_t)(p + k + shift) & RING_MASK] = ob[src + k];
                    } else dep = true;
                } else dep = true;
            }
            uint64_t mask = Gr::ballot(dep);
            lds_sync();
            while (mask) {
                int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                uint32_t w0 = L.queue[2 * j], w1 = L.queue[2 * j + 1];
                uint32_t type = w0 & 3, len = (w0 >> 2) & 0x3FFF;
                uint64_t p = pos0 + (w0 >> 16);
                if (type == T_RAW) {
                    const uint8_t *s = in + un.in_off + w1;
                    for (uint32_t k = gl; k < len; k += G) L.ring[(uint32_t)(p + k + shift) & RING_MASK] = s[k];
                } else {
                    uint32_t dist = w1;
                    int64_t src = (int64_t)p - dist;
                    for (uint32_t k = gl; k < len; k += G) {
                        int64_t s = src + (dist >= len ? k : k % dist);
                        uint8_t b;
                        if (s < 0) b = prefix_end[s];
                        else if (s >= ring_lo) b = L.ring[(uint32_t)(s + shift) & RING_MASK];
                        else b = ob[s];
                        L.ring[(uint32_t)(p + k + shift) & RING_MASK] = b;
                    }
                }
                lds_sync();
            }
        }
        pos = pos1;
        lds_sync();

        // ---- flush the ring ----
        if (alive) {
            bool last = ev == EV_DONE;
            uint32_t head = (16 - shift) & 15;
            if (flushed == 0 && head && (pos >= head || last)) {
                uint32_t n = pos < head ? (uint32_t)pos : head;
                if ((uint32_t)gl < n) ob[gl] = L.ring[(uint32_t)(gl + shift) & RING_MASK];
                flushed = n;
            }
            if (flushed || !head) {
                while (pos - flushed >= 16u * G) {
                    uint64_t a = flushed + 16u * gl;
                    uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                    *reinterpret_cast<uint4 *>(ob + a) = v;
                    flushed += 16u * G;
                }
                if (last) {
                    uint32_t pieces = (uint32_t)((pos - flushed) >> 4);
                    if ((uint32_t)gl < pieces) {
                        uint64_t a = flushed + 16u * gl;
                        uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                        *reinterpret_cast<uint4 *>(ob + a) = v;
                    }
                    flushed += 16ull * pieces;
                    uint32_t rest = (uint32_t)(pos - flushed);
                    if ((uint32_t)gl < rest) ob[flushed + gl] = L.ring[(uint32_t)(flushed + gl + shift) & RING_MASK];
                    flushed = pos;
                }
            }
        }

        // ---- table construction for the next block ----
        if (alive && (ev == EV_DYN || ev == EV_FIXED)) {
            int tstatus = SEMZ_OK;
            long overread = 0;
            if (ev == EV_DYN) {
                // code-length tree, then the code lengths themselves (deflate.lisp:176-197)
                build_tree<G, CLEN_ROOT_BITS>(L.clens, 19, L.clen, L.clen_sorted, L.clen_root, clen_entry);
                tstatus = L.clen.status;
                {   // the code lengths take at most 320 * 14 bits: make sure they are staged
                    uint64_t wp = Gr::bcast(br.wpos);
                    stage_input<G>(L.inring, in_v0, amis, limit_v, wp, s_hi);
                }
                if (leader) {
                    overread = (final_block ? 0 : MIN_BLOCK_BITS) + trailing_bits;    // deflate.lisp:392-396
                    if (tstatus == SEMZ_OK) {
                        bool cfull = overread >= (long)(L.clen.maxlen - L.clen.minlen);
                        int cmax = (int)L.clen.maxlen, total = nlit + ndist, i = 0;
                        while (i < total) {
                            br.refill();
                            uint32_t ce = L.clen_root[br.peek(CLEN_ROOT_BITS)];
                            int cl = ce & 7;
                            if (br.short_of(cfull ? cmax : cl)) { st.set(SEMZ_EOF, 0); break; }
                            br.drop(cl);
                            uint32_t sym = ce >> 3;
                            if (sym < 16) { L.lens[i++] = (uint8_t)sym; continue; }
                            int xb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                            if (br.short_of(xb)) { st.set(SEMZ_EOF, 0); break; }
                            int n = (sym == 18 ? 11 : 3) + (int)br.peek(xb);
                            br.drop(xb);
                            if (i + n > total) { st.set(SEMZ_E_DEFLATE_CODELEN_BOUNDS, 0); break; }
                            uint8_t v = 0;
                            if (sym == 16) {
                                if (i == 0) { st.set(SEMZ_E_DEFLATE_REPEAT_FIRST, 0); break; }
                                v = L.lens[i - 1];
                            }
                            for (int k = 0; k < n; k++) L.lens[i + k] = v;
                            i += n;
                        }
                    } else st.set(tstatus, 0);
                }
            } else {
                // fixed trees, deflate.lisp:146-158
                for (int i = gl; i < 320; i += G)
                    L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
                if (leader) { nlit = 288; ndist = 32; overread = final_block ? 0 : MIN_BLOCK_BITS; }
            }
            int ok = Gr::bcast((int)(st.code == SEMZ_OK));
            nlit = Gr::bcast(nlit);
            ndist = Gr::bcast(ndist);
            lds_sync();
            if (ok) {
                build_tree<G, LIT_ROOT_BITS>(L.lens, nlit, L.lit, L.lit_sorted, L.lit_root, lit_entry);
                if (L.lit.status != SEMZ_OK) { if (leader) st.set((int)L.lit.status, 0); }
                else {
                    // lengths->dist-ht special cases, deflate.lisp:91-122
                    int mode = MODE_NORMAL;
                    const uint8_t *dl = L.lens + nlit;
                    if (leader) {
                        if (ndist == 1 && dl[0] == 0) mode = MODE_LITONLY;
                        else {
                            int one = -1; bool singular = true;
                            for (int i = 0; i < ndist && singular; i++) {
                                if (dl[i] == 0) continue;
                                if (one >= 0 || dl[i] != 1) singular = false; else one = i;
                            }
                            if (singular && one >= 0) mode = MODE_SINGLE | (one << 8);
                        }
                    }
                    mode = Gr::bcast(mode);
                    if ((mode & 255) != MODE_NORMAL) {
                        // synthetic 32-entry length vector, reusing the lit/len part of lens[] (already consumed)
                        for (int i = gl; i < 32; i += G) L.lens[i] = 0;
                        lds_sync();
                        if (leader) {
                            if ((mode & 255) == MODE_LITONLY) { L.lens[30] = 1; L.lens[31] = 1; }
                            else { int one = mode >> 8; L.lens[one] = 1; L.lens[one == 31 ? 30 : 31] = 1; }
                        }
                        lds_sync();
                        build_tree<G, DIST_ROOT_BITS>(L.lens, 32, L.dist, L.dist_sorted, L.dist_root, dist_entry);
                    } else {
                        build_tree<G, DIST_ROOT_BITS>(dl, ndist, L.dist, L.dist_sorted, L.dist_root, dist_entry);
                    }
                    if (leader) {
                        if (L.dist.status != SEMZ_OK) st.set((int)L.dist.status, 0);
                        dist_mode = mode & 255;
                        lit_max = (int)L.lit.maxlen; dist_max = (int)L.dist.maxlen;
                        lit_full = overread >= (long)(L.lit.maxlen - L.lit.minlen);
                        dist_full = overread >= (long)(L.dist.maxlen - L.dist.minlen);
                        state = ST_HUFF;
                    }
                }
            }
            lds_sync();
        }

        // ---- unit end: trailer and results (leader) ----
        if (ev == EV_DONE && alive) {
            if (leader) {
                semz_tail t{};
                semz_result r{};
                if (st.code == SEMZ_OK && finished) {
                    // the trailer is byte aligned: read it straight from memory
                    uint64_t bp = ((br.bitpos() + 7) >> 3) - amis, endb = bp;    // offset inside `in`
                    if (un.kind == SEMZ_UNIT_GZIP) {            // gzip.lisp:102-113
                        t.check_kind = SEMZ_TK_CRC32;
                        if (un.in_limit >= bp + 4) {
                            t.have_check = 1; t.want[0] = ld_u32(in + bp); endb = bp + 4;
                            if (un.in_limit >= bp + 8) { t.have_isize = 1; t.want_isize = ld_u32(in + bp + 4); endb = bp + 8; }
                        }
                    } else if (un.kind == SEMZ_UNIT_ZLIB) {     // zlib.lisp:66-76
                        t.check_kind = SEMZ_TK_ADLER32;
                        if (un.in_limit >= bp + 4) {
                            t.have_check = 1; t.want[0] = __builtin_bswap32(ld_u32(in + bp)); endb = bp + 4;
                        }
                        t.have_isize = 1; t.want_isize = (uint32_t)pos;
                    }
                    if (un.kind != SEMZ_UNIT_DEFLATE) br.init(L.inring, endb + amis, limit_v);
                }
                t.trailer_end_bit = br.bitpos() - 8 * amis;
                r.status = st.code;
                r.detail = st.detail;
                r.in_end_bit = br.bitpos() - 8 * amis;
                r.out_len = pos;
                r.out_off = un.out_off;
                r.aux = aux;
                results[u] = r;
                tails[u] = t;
            }
            alive = false;
        }
    }
}

} // namespace

extern "C" void semz_launch_inflate(int group, const uint8_t *d_in, const semz_unit *d_units, uint32_t nunits,
                                    uint8_t *d_out, semz_result *d_results, semz_tail *d_tails,
                                    hipStream_t stream)
{
    if (nunits == 0) return;
    switch (group) {
    case 16: inflate_kernel<16><<<(nunits + 3) / 4, 64, 0, stream>>>(d_in, d_units, nunits, d_out, d_results, d_tails); break;
    case 32: inflate_kernel<32><<<(nunits + 1) / 2, 64, 0, stream>>>(d_in, d_units, nunits, d_out, d_results, d_tails); break;
    default: inflate_kernel<64><<<nunits, 64, 0, stream>>>(d_in, d_units, nunits, d_out, d_results, d_tails); break;
    }
}
