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
// Layout: one unit per 64-lane wavefront, one wavefront per workgroup, 16+
// workgroups per CU (about 8 KiB of LDS each).
//
// The bit-serial part of a unit is wave-uniform: the 64-bit bit buffer, the
// block state machine and every branch live in scalar registers (values read
// from LDS come back through v_readfirstlane, the compressed input is fetched
// with scalar loads one dword ahead of its use), so the vector ALU is left to
// the data-parallel parts.  Decoding proceeds in rounds of up to 64 tokens;
// token t of a round is parked in the vector registers of lane t.  A round is
// then resolved by all 64 lanes:
//   literals        one ds_write_b8 for all literal lanes
//   far matches     (source older than the LDS ring) gathered from HBM four
//                   matches at a time so that their latencies overlap
//   near matches    in token order, one wave-wide LDS copy each (periodic
//                   replication when distance < length, common.lisp:324-333);
//                   LDS operations of a wave execute in order, so consecutive
//                   copies need no barrier
// Output is staged in a 4 KiB LDS ring whose index equals the HBM address
// modulo 16 and flushed with 16-byte stores, 1 KiB per wave instruction.
// Huffman tables are rebuilt in LDS per block by all 64 lanes: a root lookup
// table plus a canonical (first-code / sorted-symbol) description used for
// codes longer than the root and for the table fill itself.
#include "semz_device.h"

namespace {

constexpr int LIT_ROOT_BITS = 10;
constexpr int DIST_ROOT_BITS = 8;
constexpr int CLEN_ROOT_BITS = 7;
constexpr int RING = 4096;               // output staging ring per unit (bytes)
constexpr int RING_MASK = RING - 1;
constexpr int ROUND_BYTES = 2048;        // a round stops once it has produced this many bytes
constexpr int FLUSH_BYTES = 1024;        // 64 lanes x 16 bytes
constexpr int MIN_BLOCK_BITS = 10;       // +min-deflate-block-bitsize+, deflate.lisp:4-12
static_assert(RING - ROUND_BYTES - 258 >= FLUSH_BYTES + 258, "far sources must already be flushed");

enum { T_LIT = 0, T_MATCH = 1, T_RAW = 2 };
enum { EV_ROUND = 0, EV_DYN = 1, EV_FIXED = 2, EV_DONE = 3 };
enum { ST_HDR = 0, ST_HUFF = 1, ST_STORED = 2 };
enum { MODE_NORMAL = 0, MODE_LITONLY = 1, MODE_SINGLE = 2 };

// lit/len root entry (16 bit): [3:0] code length (0 = longer than root) [5:4] kind [14:6] payload
//   literal: byte; length: symbol - 257 (0..28); end of block: -; invalid 286/287: symbol - 257
// dist root entry (16 bit):    [3:0] code length, [4] invalid (30/31), [9:5] code
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_BAD = 3 };

typedef const __attribute__((address_space(4))) uint32_t *k_u32p;   // scalar-load view of read-only memory

template <typename T> __device__ __forceinline__ T uni(T v)
{
    if constexpr (sizeof(T) == 8) {
        uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
        return (T)(((uint64_t)hi << 32) | lo);
    } else {
        return (T)__builtin_amdgcn_readfirstlane((int)v);
    }
}

struct TreeMeta {
    uint16_t count[16];
    uint16_t first[16];   // first canonical code (MSB-first) per length
    uint16_t offs[16];    // start of each length's run in sorted[]
    uint32_t limit[16];   // (first+count) << (15-len): exclusive bound of the length, left aligned to 15 bits
    uint32_t minlen, maxlen, status, pad;
};

struct __align__(16) UnitLds {
    uint8_t ring[RING];
    uint16_t lit_root[1 << LIT_ROOT_BITS];
    uint16_t dist_root[1 << DIST_ROOT_BITS];
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
    if (sym < 256) return (sym << 6) | (K_LIT << 4) | len;
    if (sym == 256) return (K_EOB << 4) | len;
    if (sym > 285) return ((sym - 257) << 6) | (K_BAD << 4) | len;
    return ((sym - 257) << 6) | (K_LEN << 4) | len;
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t code, uint32_t len)
{
    return (code << 5) | (code > 29 ? 1u << 4 : 0u) | len;
}
__device__ __forceinline__ uint32_t clen_entry(uint32_t sym, uint32_t len) { return (sym << 3) | len; }

// base length and extra-bit count of length symbol 257+i (deflate.lisp:14-23)
__device__ __forceinline__ void length_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    if (i < 8) { extra = 0; base = 3 + i; }
    else if (i == 28) { extra = 0; base = 258; }
    else { extra = (i >> 2) - 1; base = 3 + ((4 + (i & 3)) << extra); }
}
// base distance and extra-bit count of distance code c (deflate.lisp:25-34)
__device__ __forceinline__ void distance_code(uint32_t c, uint32_t &base, uint32_t &extra)
{
    if (c < 4) { extra = 0; base = c + 1; }
    else { extra = (c >> 1) - 1; base = 1 + ((2 + (c & 1)) << extra); }
}

// Canonical decode of a left-aligned 15-bit MSB-first window, lengths lo..hi.
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

// Wave-parallel construction of one canonical tree (lengths->ht,
// huffman.lisp:142-225): per-length counts (one lane per length), the
// exact-completeness check (huffman.lisp:166-184), the sorted symbol list and
// the root table.
template <int ROOT_BITS, typename SortT, typename RootT, typename MakeEntry>
__device__ void build_tree(const uint8_t *lens, int n, TreeMeta &m, SortT *sorted, RootT *root,
                           MakeEntry make_entry)
{
    const int lane = threadIdx.x & 63;
    // 1. counts
    uint32_t c = 0;
    if (lane >= 1 && lane < 16)
        for (int s = 0; s < n; s++) c += (lens[s] == lane);
    if (lane < 16) m.count[lane] = (uint16_t)c;
    lds_sync();
    // 2. canonical bookkeeping
    if (lane == 0) {
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
    if (lane >= 1 && lane < 16) {
        uint32_t k = m.offs[lane];
        for (int s = 0; s < n; s++)
            if (lens[s] == lane) sorted[k++] = (SortT)s;
    }
    lds_sync();
    // 4. root table, one index per lane per step
    for (int i = lane; i < (1 << ROOT_BITS); i += 64) {
        uint32_t v15 = (__brev((uint32_t)i) >> (32 - ROOT_BITS)) << (15 - ROOT_BITS);
        uint32_t idx = 0;
        uint32_t l = canon_find(m, v15, 1, ROOT_BITS, idx);
        root[i] = l ? (RootT)make_entry((uint32_t)sorted[idx], l) : (RootT)0;
    }
    lds_sync();
}

// LSB-first bit reader (lsb-bit-reader, bits.lisp:118), wave-uniform.
// Positions are "v-offsets": byte offset inside `in` plus the 4-byte
// misalignment of the `in` pointer, so that the input can be fetched with
// aligned scalar dword loads.  The dword after the last one consumed is always
// in flight (nextw), which hides the load latency behind the symbols decoded
// from the current buffer.  Bytes at or beyond `limit` read as zero
// (bits.lisp:49-50); near_end turns on the exact end-of-input accounting that
// decides `eof` (bits.lisp:35-45).
struct BitReader {
    k_u32p base4;     // base4[v >> 2] is the dword at v-offset v
    uint64_t wpos;    // v-offset of the dword held in nextw (multiple of 4)
    uint64_t limit;   // v-offset of the first byte that does not exist
    uint64_t buf;
    uint32_t nextw;
    int cnt;          // bits in buf
    int skip;         // bits to discard from the first dword (unaligned start)
    bool near_end;

    __device__ __forceinline__ uint32_t load_word(uint64_t v) const
    {
        if (v >= limit) return 0;
        uint32_t w = base4[v >> 2];
        if (v + 4 > limit) w &= (1u << (8 * (uint32_t)(limit - v))) - 1u;
        return w;
    }
    __device__ __forceinline__ void init(k_u32p b4, uint64_t v, uint64_t lim)
    {
        base4 = b4; wpos = v & ~3ull; limit = lim; buf = 0; cnt = 0; skip = (int)(v & 3) * 8; near_end = false;
        nextw = load_word(wpos);
    }
    __device__ __forceinline__ uint32_t take()
    {
        uint32_t w = nextw;
        if (wpos + 4 > limit) near_end = true;
        wpos += 4;
        nextw = load_word(wpos);
        return w;
    }
    // after refill(): cnt >= 33
    __device__ __forceinline__ void refill()
    {
        if (skip) {
            buf = (uint64_t)take() >> skip;
            cnt = 32 - skip;
            skip = 0;
        }
        if (cnt <= 32) {
            buf |= (uint64_t)take() << cnt;
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

__device__ __forceinline__ uint32_t ld_bytes_le(const uint8_t *p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// one byte of history at output position s (s < pos0): preset dictionary, LDS ring or HBM
__device__ __forceinline__ uint32_t history_byte(int64_t s, int64_t near_lo, const uint8_t *ring, uint32_t shift,
                                                 uint8_t *ob, const uint8_t *prefix_end)
{
    if (s >= near_lo) return ring[(uint32_t)((uint64_t)s + shift) & RING_MASK];
    if (s < 0) return prefix_end[s];
    // written by this wave earlier; read it from L2, not from a possibly stale L1 line
    return __hip_atomic_load(ob + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64, 4) void inflate_kernel(const uint8_t *in, const semz_unit *units,
                                                     uint32_t nunits, uint8_t *out,
                                                     semz_result *results, semz_tail *tails)
{
    __shared__ UnitLds L;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;

    // the unit descriptor through scalar loads (wave-uniform)
    semz_unit un;
    {
        k_u32p up = (k_u32p)(uintptr_t)(units + u);
        uint32_t w[sizeof(semz_unit) / 4];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(semz_unit) / 4); i++) w[i] = up[i];
        __builtin_memcpy(&un, w, sizeof un);
    }

    uint8_t *ob = out + un.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);      // ring index == global address (mod 16)
    const uint64_t out_cap = un.out_cap;

    // v-offsets (see BitReader): offset inside `in` + misalignment of the `in` pointer
    const uint64_t amis = (uint64_t)((uintptr_t)in & 3);
    const k_u32p base4 = (k_u32p)((uintptr_t)in - amis);
    const uint64_t limit_v = un.in_limit + amis;
    BitReader br;
    Status st{SEMZ_OK, 0};
    br.init(base4, un.in_off + amis, limit_v);
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
    const int trailing_bits = un.kind == SEMZ_UNIT_GZIP ? 64 : un.kind == SEMZ_UNIT_ZLIB ? 32 : 0;

    uint64_t pos = 0;                    // output bytes produced
    uint64_t flushed = 0;                // output bytes already in HBM

    // ---- container header ----
    if (un.kind == SEMZ_UNIT_GZIP) {
        Status hs{SEMZ_OK, 0};
        uint64_t p = parse_gzip_header(in, un.in_off, un.in_limit, hs);
        p = uni(p);
        st.code = uni(hs.code);
        st.detail = uni(hs.detail);
        aux = (uint32_t)(p - un.in_off);
        br.init(base4, p + amis, limit_v);
        wsize = 32768u;
    } else if (un.kind == SEMZ_UNIT_ZLIB) {          // zlib.lisp:21-57
        uint64_t p = un.in_off;
        if (un.in_limit - p < 2) st.set(SEMZ_EOF, 0);
        else {
            uint32_t cmf = uni((uint32_t)in[p]), flg = uni((uint32_t)in[p + 1]);
            p += 2;
            if ((cmf & 15) != 8) st.set(SEMZ_E_ZLIB_METHOD, cmf & 15);
            else if ((cmf >> 4) > 7) st.set(SEMZ_E_ZLIB_CINFO, cmf >> 4);
            else if ((cmf * 256 + flg) % 31) st.set(SEMZ_E_ZLIB_FCHECK, 0);
            else {
                wsize = 1u << ((cmf >> 4) + 8);
                if (flg & 32) {
                    if (un.in_limit - p < 4) st.set(SEMZ_EOF, 0);
                    else {
                        aux = uni(__builtin_bswap32(ld_bytes_le(in + p, 4)));
                        p += 4;
                        if (!(un.flags & SEMZ_UF_HAVE_DICT_FN)) st.set(SEMZ_E_ZLIB_NO_DICT_FN, 0);
                        else if (un.prefix_len == 0 && un.prefix_off == 0) st.set(SEMZ_E_ZLIB_UNKNOWN_DICT, aux);
                    }
                }
            }
            br.init(base4, p + amis, limit_v);
        }
    }
    if (un.prefix_len) {                              // deflate.lisp:419-425
        prefix_amount = un.prefix_len < wsize ? un.prefix_len : wsize;
        prefix_end = in + un.prefix_off + un.prefix_len;
    }

    // ---- main loop: one round per iteration ----
    for (;;) {
        int ev = st.code != SEMZ_OK ? EV_DONE : EV_ROUND;
        uint32_t ntok = 0, rbytes = 0;
        uint32_t tk0 = 0, tk1 = 0;                   // token `lane` of this round
#define EMIT(w0, w1) do { if (lane == ntok) { tk0 = (w0); tk1 = (w1); } ntok++; } while (0)
#define FAIL(c, d) { st.set((c), (d)); ev = EV_DONE; break; }
        while (ev == EV_ROUND && ntok < 64u && rbytes < (uint32_t)ROUND_BYTES) {
            br.refill();
            if (state == ST_HUFF) {
                // ---- one literal/length symbol, decode-huffman-data deflate.lisp:313-338 ----
                uint32_t e = uni((uint32_t)L.lit_root[br.peek(LIT_ROOT_BITS)]);
                uint32_t cl = e & 15;
                if (cl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    cl = uni(canon_find(L.lit, v15, LIT_ROOT_BITS + 1, 15, idx));
                    e = lit_entry(uni((uint32_t)L.lit_sorted[uni(idx)]), cl);
                }
                if (br.short_of(lit_full ? lit_max : (int)cl)) FAIL(SEMZ_EOF, 0)
                br.drop(cl);
                const uint32_t kind = (e >> 4) & 3, payload = e >> 6;
                if (kind == K_LIT) {
                    if (pos + rbytes >= out_cap) FAIL(SEMZ_E_OUT_CAP, 0)
                    EMIT(T_LIT | (1u << 2) | (rbytes << 16), payload);
                    rbytes++;
                    continue;
                }
                if (kind == K_EOB) { state = ST_HDR; continue; }
                if (kind == K_BAD) FAIL(SEMZ_E_DEFLATE_LITLEN_CODE, 257 + payload)
                // length + distance
                if (dist_mode == MODE_LITONLY) FAIL(SEMZ_E_DEFLATE_LEN_IN_LITONLY, 257 + payload)
                uint32_t lbase, lx;
                length_code(payload, lbase, lx);
                if (br.short_of(lx)) FAIL(SEMZ_EOF, 0)
                const uint32_t length = lbase + br.peek(lx);
                br.drop(lx);
                br.refill();
                uint32_t d = uni((uint32_t)L.dist_root[br.peek(DIST_ROOT_BITS)]);
                uint32_t dl = d & 15;
                if (dl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    dl = uni(canon_find(L.dist, v15, DIST_ROOT_BITS + 1, 15, idx));
                    d = dist_entry(uni((uint32_t)L.dist_sorted[uni(idx)]), dl);
                }
                if (br.short_of(dist_full ? dist_max : (int)dl)) FAIL(SEMZ_EOF, 0)
                br.drop(dl);
                if (d & 16) FAIL(dist_mode == MODE_SINGLE ? SEMZ_E_DEFLATE_SINGLE_DIST : SEMZ_E_DEFLATE_DIST_CODE, (d >> 5) & 31)
                uint32_t dbase, dx;
                distance_code((d >> 5) & 31, dbase, dx);
                if (br.short_of(dx)) FAIL(SEMZ_EOF, 0)
                const uint32_t distance = dbase + br.peek(dx);
                br.drop(dx);
                const uint64_t pointer = prefix_amount + pos + rbytes;          // deflate.lisp:329-331
                const uint64_t lim = pointer < wsize ? pointer : wsize;
                if (distance > lim) FAIL(SEMZ_E_DEFLATE_DISTANCE, distance)
                if (pos + rbytes + length > out_cap) FAIL(SEMZ_E_OUT_CAP, 0)
                EMIT(T_MATCH | (length << 2) | (rbytes << 16), distance);
                rbytes += length;
                continue;
            }
            if (state == ST_HDR) {
                if (final_block) { finished = true; ev = EV_DONE; break; }
                // BFINAL, BTYPE: deflate.lisp:374-397
                if (br.short_of(1)) FAIL(SEMZ_EOF, 0)
                final_block = br.peek(1); br.drop(1);
                if (br.short_of(2)) FAIL(SEMZ_EOF, 0)
                const uint32_t type = br.peek(2); br.drop(2);
                if (type == 0) {
                    br.align_byte();
                    br.refill();
                    if (br.short_of(16)) FAIL(SEMZ_EOF, 0)
                    const uint32_t len = br.peek(16); br.drop(16);
                    br.refill();
                    if (br.short_of(16)) FAIL(SEMZ_EOF, 0)
                    const uint32_t chk = br.peek(16); br.drop(16);
                    if (chk != (len ^ 0xFFFFu)) FAIL(SEMZ_E_DEFLATE_STORED_LEN, chk)
                    stored_left = len;
                    if (len) {
                        state = ST_STORED;
                        // the whole stored block must exist (decode-uncompressed-data, deflate.lisp:350)
                        if (pos + rbytes + len > out_cap) FAIL(SEMZ_E_OUT_CAP, 0)
                        if (br.remaining() < (int64_t)8 * len) FAIL(SEMZ_EOF, 0)
                    }
                } else if (type == 1) { ev = EV_FIXED; break; }
                else if (type == 2) {
                    // read-dynamic-dhi, deflate.lisp:160-175: counts and code-length code lengths
                    if (br.short_of(5)) FAIL(SEMZ_EOF, 0)
                    nlit = 257 + br.peek(5); br.drop(5);
                    if (br.short_of(5)) FAIL(SEMZ_EOF, 0)
                    ndist = 1 + br.peek(5); br.drop(5);
                    if (br.short_of(4)) FAIL(SEMZ_EOF, 0)
                    const int nclen = 4 + br.peek(4); br.drop(4);
                    if (lane < 19) L.clens[lane] = 0;
                    lds_sync();
                    bool eof = false;
                    for (int i = 0; i < nclen; i++) {
                        // code-length order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, one nibble/5 bits each
                        const uint64_t ORDER_LO = 0x0B050A0609070800ull;   // entries 3..10 (bytes)
                        uint32_t sym;
                        if (i < 3) sym = 16 + i;
                        else if (i < 11) sym = (uint32_t)(ORDER_LO >> (8 * (i - 3))) & 0xFF;
                        else sym = (i & 1) ? (uint32_t)(19 - i) / 2u : (uint32_t)(i + 13) / 2u;
                        br.refill();
                        if (br.short_of(3)) { eof = true; break; }
                        if (lane == 0) L.clens[sym] = (uint8_t)br.peek(3);
                        br.drop(3);
                    }
                    if (eof) FAIL(SEMZ_EOF, 0)
                    ev = EV_DYN;
                    break;
                } else FAIL(SEMZ_E_DEFLATE_BTYPE, 0)
                continue;
            }
            // ---- ST_STORED: raw bytes; the bit buffer is byte aligned, hand the span to the wave ----
            {
                const uint32_t room = ROUND_BYTES - rbytes;
                const uint32_t n = stored_left < room ? stored_left : room;
                const uint64_t src = br.bitpos() >> 3;            // v-offset of the span
                EMIT(T_RAW | (n << 2) | (rbytes << 16), (uint32_t)(src - amis - un.in_off));
                rbytes += n;
                stored_left -= n;
                br.init(base4, src + n, limit_v);                 // restart the bit buffer after the span
                if (stored_left == 0) state = ST_HDR;
            }
        }
#undef EMIT
#undef FAIL

        // ---- resolve the round's tokens ----
        const uint64_t pos0 = pos, pos1 = pos + rbytes;
        // positions >= near_lo live in the ring; older ones are in HBM (flushed), negative ones in the preset dictionary
        const int64_t near_lo = (int64_t)pos1 > RING ? (int64_t)pos1 - RING : 0;
        if (ntok) {
            const bool valid = lane < ntok;
            const uint32_t type = tk0 & 3, mylen = (tk0 >> 2) & 0x3FFF;
            const uint64_t myp = pos0 + (tk0 >> 16);
            if (valid && type == T_LIT) L.ring[(uint32_t)(myp + shift) & RING_MASK] = (uint8_t)tk1;
            const bool is_match = valid && type == T_MATCH;
            const bool far = is_match && ((int64_t)myp - (int64_t)tk1) < near_lo;
            uint64_t farm = __ballot(far);
            uint64_t nearm = __ballot(is_match && !far);
            uint64_t rawm = __ballot(valid && type == T_RAW);
            (void)mylen;
            // far matches: the source predates the ring (or lies in the preset dictionary); no byte of
            // it was produced in this round, so up to four gathers are put in flight together
            while (farm) {
                uint32_t fl[4], fp[4], fv[4];
                int64_t fs[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    fl[b] = 0; fp[b] = 0; fs[b] = 0; fv[b] = 0;
                    if (farm) {
                        const int j = __builtin_ctzll(farm);
                        farm &= farm - 1;
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)tk0, j);
                        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)tk1, j);
                        fl[b] = (w0 >> 2) & 0x3FFF;
                        fp[b] = w0 >> 16;
                        fs[b] = (int64_t)(pos0 + fp[b]) - (int64_t)w1;
                        if (lane < fl[b]) fv[b] = history_byte(fs[b] + lane, near_lo, L.ring, shift, ob, prefix_end);
                    }
                }
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (lane < fl[b]) L.ring[(uint32_t)(pos0 + fp[b] + lane + shift) & RING_MASK] = (uint8_t)fv[b];
#pragma unroll
                for (int b = 0; b < 4; b++)
                    for (uint32_t k = 64 + lane; k < fl[b]; k += 64)
                        L.ring[(uint32_t)(pos0 + fp[b] + k + shift) & RING_MASK] =
                            (uint8_t)history_byte(fs[b] + k, near_lo, L.ring, shift, ob, prefix_end);
            }
            // stored spans straight from the input
            while (rawm) {
                const int j = __builtin_ctzll(rawm);
                rawm &= rawm - 1;
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)tk0, j);
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)tk1, j);
                const uint32_t len = (w0 >> 2) & 0x3FFF;
                const uint64_t p = pos0 + (w0 >> 16);
                const uint8_t *s = in + un.in_off + w1;
                for (uint32_t k = lane; k < len; k += 64) L.ring[(uint32_t)(p + k + shift) & RING_MASK] = s[k];
            }
            // near matches in token order; LDS operations of one wave execute in order
            while (nearm) {
                const int j = __builtin_ctzll(nearm);
                nearm &= nearm - 1;
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)tk0, j);
                const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)tk1, j);
                const uint32_t len = (w0 >> 2) & 0x3FFF;
                const uint32_t p = (uint32_t)(pos0 + (w0 >> 16)) + shift;     // ring index of the destination
                const uint32_t src = p - dist;
                if (dist >= len) {
                    for (uint32_t k = lane; k < len; k += 64)
                        L.ring[(p + k) & RING_MASK] = L.ring[(src + k) & RING_MASK];
                } else if (dist == 1) {
                    const uint8_t b = L.ring[src & RING_MASK];
                    for (uint32_t k = lane; k < len; k += 64) L.ring[(p + k) & RING_MASK] = b;
                } else {
                    for (uint32_t k = lane; k < len; k += 64)
                        L.ring[(p + k) & RING_MASK] = L.ring[(src + k % dist) & RING_MASK];
                }
            }
        }
        pos = pos1;
        lds_sync();

        // ---- flush the ring ----
        {
            const bool last = ev == EV_DONE;
            const uint32_t head = (16 - shift) & 15;
            if (flushed == 0 && head && (pos >= head || last)) {
                const uint32_t n = pos < head ? (uint32_t)pos : head;
                if (lane < n) ob[lane] = L.ring[(lane + shift) & RING_MASK];
                flushed = n;
            }
            if (flushed || !head) {
                while (pos - flushed >= (uint64_t)FLUSH_BYTES) {
                    const uint64_t a = flushed + 16u * lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                    *reinterpret_cast<uint4 *>(ob + a) = v;
                    flushed += FLUSH_BYTES;
                }
                if (last) {
                    const uint32_t pieces = (uint32_t)((pos - flushed) >> 4);
                    if (lane < pieces) {
                        const uint64_t a = flushed + 16u * lane;
                        const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                        *reinterpret_cast<uint4 *>(ob + a) = v;
                    }
                    flushed += 16ull * pieces;
                    const uint32_t rest = (uint32_t)(pos - flushed);
                    if (lane < rest) ob[flushed + lane] = L.ring[(uint32_t)(flushed + lane + shift) & RING_MASK];
                    flushed = pos;
                }
            }
        }

        // ---- table construction for the next block ----
        if (ev == EV_DYN || ev == EV_FIXED) {
            long overread = 0;
            if (ev == EV_DYN) {
                // code-length tree, then the code lengths themselves (deflate.lisp:176-197)
                build_tree<CLEN_ROOT_BITS>(L.clens, 19, L.clen, L.clen_sorted, L.clen_root, clen_entry);
                const int tstatus = uni((int)L.clen.status);
                overread = (final_block ? 0 : MIN_BLOCK_BITS) + trailing_bits;    // deflate.lisp:392-396
                if (tstatus == SEMZ_OK) {
                    const int cmaxl = uni((int)L.clen.maxlen), cminl = uni((int)L.clen.minlen);
                    const bool cfull = overread >= (long)(cmaxl - cminl);
                    const int total = nlit + ndist;
                    int i = 0;
                    while (i < total) {
                        br.refill();
                        const uint32_t ce = uni((uint32_t)L.clen_root[br.peek(CLEN_ROOT_BITS)]);
                        const int cl = ce & 7;
                        if (br.short_of(cfull ? cmaxl : cl)) { st.set(SEMZ_EOF, 0); break; }
                        br.drop(cl);
                        const uint32_t sym = ce >> 3;
                        if (sym < 16) { if (lane == 0) L.lens[i] = (uint8_t)sym; i++; continue; }
                        const int xb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                        if (br.short_of(xb)) { st.set(SEMZ_EOF, 0); break; }
                        const int n = (sym == 18 ? 11 : 3) + (int)br.peek(xb);
                        br.drop(xb);
                        if (i + n > total) { st.set(SEMZ_E_DEFLATE_CODELEN_BOUNDS, 0); break; }
                        uint32_t v = 0;
                        if (sym == 16) {
                            if (i == 0) { st.set(SEMZ_E_DEFLATE_REPEAT_FIRST, 0); break; }
                            lds_sync();
                            v = L.lens[i - 1];
                        }
                        if (lane < (uint32_t)n) L.lens[i + lane] = (uint8_t)v;
                        if (lane + 64 < (uint32_t)n) L.lens[i + lane + 64] = (uint8_t)v;
                        if (lane + 128 < (uint32_t)n) L.lens[i + lane + 128] = (uint8_t)v;
                        i += n;
                    }
                } else st.set(tstatus, 0);
            } else {
                // fixed trees, deflate.lisp:146-158
                for (int i = lane; i < 320; i += 64)
                    L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
                nlit = 288; ndist = 32; overread = final_block ? 0 : MIN_BLOCK_BITS;
            }
            lds_sync();
            if (st.code == SEMZ_OK) {
                build_tree<LIT_ROOT_BITS>(L.lens, nlit, L.lit, L.lit_sorted, L.lit_root, lit_entry);
                const int ls = uni((int)L.lit.status);
                if (ls != SEMZ_OK) st.set(ls, 0);
                else {
                    // lengths->dist-ht special cases, deflate.lisp:91-122
                    int mode = MODE_NORMAL;
                    const uint8_t *dl = L.lens + nlit;
                    if (ndist == 1 && dl[0] == 0) mode = MODE_LITONLY;
                    else {
                        int one = -1; bool singular = true;
                        for (int i = 0; i < ndist && singular; i++) {
                            if (dl[i] == 0) continue;
                            if (one >= 0 || dl[i] != 1) singular = false; else one = i;
                        }
                        if (singular && one >= 0) mode = MODE_SINGLE | (one << 8);
                    }
                    mode = uni(mode);
                    if ((mode & 255) != MODE_NORMAL) {
                        // synthetic 32-entry length vector, reusing the lit/len part of lens[] (already consumed)
                        lds_sync();
                        if (lane < 32) {
                            uint8_t v = 0;
                            if ((mode & 255) == MODE_LITONLY) v = lane >= 30;
                            else { const uint32_t one = mode >> 8; v = lane == one || lane == (one == 31 ? 30u : 31u); }
                            L.lens[lane] = v;
                        }
                        lds_sync();
                        build_tree<DIST_ROOT_BITS>(L.lens, 32, L.dist, L.dist_sorted, L.dist_root, dist_entry);
                    } else {
                        build_tree<DIST_ROOT_BITS>(dl, ndist, L.dist, L.dist_sorted, L.dist_root, dist_entry);
                    }
                    const int ds = uni((int)L.dist.status);
                    if (ds != SEMZ_OK) st.set(ds, 0);
                    dist_mode = mode & 255;
                    lit_max = uni((int)L.lit.maxlen); dist_max = uni((int)L.dist.maxlen);
                    lit_full = overread >= (long)(lit_max - uni((int)L.lit.minlen));
                    dist_full = overread >= (long)(dist_max - uni((int)L.dist.minlen));
                    state = ST_HUFF;
                }
            }
            lds_sync();
        }

        // ---- unit end: trailer and results ----
        if (ev == EV_DONE) {
            semz_tail t{};
            semz_result r{};
            if (st.code == SEMZ_OK && finished) {
                // the trailer is byte aligned: read it straight from memory
                const uint64_t bp = ((br.bitpos() + 7) >> 3) - amis;            // offset inside `in`
                uint64_t endb = bp;
                if (un.kind == SEMZ_UNIT_GZIP) {            // gzip.lisp:102-113
                    t.check_kind = SEMZ_TK_CRC32;
                    if (un.in_limit >= bp + 4) {
                        t.have_check = 1; t.want[0] = ld_bytes_le(in + bp, 4); endb = bp + 4;
                        if (un.in_limit >= bp + 8) { t.have_isize = 1; t.want_isize = ld_bytes_le(in + bp + 4, 4); endb = bp + 8; }
                    }
                } else if (un.kind == SEMZ_UNIT_ZLIB) {     // zlib.lisp:66-76
                    t.check_kind = SEMZ_TK_ADLER32;
                    if (un.in_limit >= bp + 4) {
                        t.have_check = 1; t.want[0] = __builtin_bswap32(ld_bytes_le(in + bp, 4)); endb = bp + 4;
                    }
                    t.have_isize = 1; t.want_isize = (uint32_t)pos;
                }
                if (un.kind != SEMZ_UNIT_DEFLATE) br.init(base4, endb + amis, limit_v);
            }
            t.trailer_end_bit = br.bitpos() - 8 * amis;
            r.status = st.code;
            r.detail = st.detail;
            r.in_end_bit = br.bitpos() - 8 * amis;
            r.out_len = pos;
            r.out_off = un.out_off;
            r.aux = aux;
            if (lane == 0) {
                results[u] = r;
                tails[u] = t;
            }
            break;
        }
    }
}

} // namespace

extern "C" void semz_launch_inflate(const uint8_t *d_in, const semz_unit *d_units, uint32_t nunits,
                                    uint8_t *d_out, semz_result *d_results, semz_tail *d_tails,
                                    hipStream_t stream)
{
    if (nunits == 0) return;
    inflate_kernel<<<nunits, 64, 0, stream>>>(d_in, d_units, nunits, d_out, d_results, d_tails);
}
