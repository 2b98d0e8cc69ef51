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
// Layout: one 64-lane wavefront per workgroup, four units per wavefront
// (16 lanes each), one wavefront per SIMD (about 34 KiB of LDS per workgroup).
// The bit-serial Huffman decode is bound by instruction issue, not by memory:
// four group leaders (lane 0 of each group) decode in lock step, so every
// vector instruction of the serial loop advances four units, and a SIMD that
// hosts a single wave issues it back to back.  (A wave-uniform scalar decode,
// one unit per wave, was measured first: 16 waves of a CU then queue on the one
// scalar issue slot per SIMD and a token costs about 2,200 cycles.)
//
// Decoding proceeds in rounds.  Within a round a leader writes literals
// straight into the unit's LDS output ring and queues up to 32 match / stored
// tokens in LDS; the 16 lanes of the group then resolve the queue:
//   far matches     (source older than the ring, or in the preset dictionary)
//                   gathered from HBM, four tokens x 32 bytes in flight
//   stored spans    copied from the input
//   near matches    in token order, one group-wide LDS copy each (periodic
//                   replication when distance < length, common.lisp:324-333);
//                   LDS operations of a wave execute in order, so consecutive
//                   copies need no barrier
// The ring index equals the HBM address modulo 16 and the ring is flushed with
// 16-byte stores.  Huffman tables are rebuilt in LDS per block by the 16 lanes:
// a root lookup table plus a canonical (first-code / sorted-symbol)
// description used for codes longer than the root and for the table fill.
#include "semz_device.h"

#ifdef SEMZ_STAMPS
// diagnostic build only (make STAMPS=1): per-phase shader-clock totals summed over all units
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL uint64_t t_last = __builtin_amdgcn_s_memtime(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_acc[4] = {0, 0, 0, 0};
#define STAMP(i) do { uint64_t t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define COUNT(i, n) do { n_acc[i] += (n); } while (0)
#else
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define COUNT(i, n) do { } while (0)
#endif

namespace {

constexpr int G = 16;                    // lanes per unit
constexpr int UPW = 64 / G;              // units per wave
constexpr int LIT_ROOT_BITS = 10;
constexpr int DIST_ROOT_BITS = 8;
constexpr int CLEN_ROOT_BITS = 7;
constexpr int RING = 4096;               // output staging ring per unit (bytes)
constexpr int RING_MASK = RING - 1;
constexpr int ROUND_BYTES = 2048;        // a round stops once it has produced this many bytes
constexpr int FLUSH_BYTES = 16 * G;      // one 16-byte store per lane
constexpr int QCAP = 32;                 // match / stored tokens per round
constexpr int IN_RING = 1024;            // compressed-input staging ring per unit (bytes)
constexpr int STAGE_BYTES = 16 * G;      // one 16-byte load per lane
constexpr int MIN_BLOCK_BITS = 10;       // +min-deflate-block-bitsize+, deflate.lisp:4-12
static_assert(RING - ROUND_BYTES - 258 >= FLUSH_BYTES + 258, "far sources must already be flushed");

using Gr = Group<G>;

enum { T_MATCH = 1, T_RAW = 2 };
enum { EV_ROUND = 0, EV_DYN = 1, EV_FIXED = 2, EV_DONE = 3 };
enum { ST_HDR = 0, ST_HUFF = 1, ST_STORED = 2 };
enum { MODE_NORMAL = 0, MODE_LITONLY = 1, MODE_SINGLE = 2 };

// lit/len root entry (16 bit): [3:0] code length (0 = longer than the root), [4] literal
//   literal:      [15:8] byte
//   non-literal:  [7:5] extra-bit count, [15:8] base length - 3;
//                 extra == 7 marks the rest: [15:8] == 0 end of block, else symbol - 257 of 286/287
// dist root entry (16 bit):    [3:0] code length, [7:4] extra-bit count, [9:8] hi, [10] invalid (30/31),
//                              [15:11] code; distance = 1 + (hi << extra) + extra bits
constexpr uint32_t LIT_FLAG = 1u << 4;
constexpr uint32_t DIST_BAD = 1u << 10;

struct TreeMeta {
    uint16_t count[16];
    uint16_t first[16];   // first canonical code (MSB-first) per length
    uint16_t offs[16];    // start of each length's run in sorted[]
    uint32_t limit[16];   // (first+count) << (15-len): exclusive bound of the length, left aligned to 15 bits
    uint32_t minlen, maxlen, status, pad;
};

struct __align__(16) UnitLds {
    uint8_t ring[RING];
    uint8_t inring[IN_RING];
    uint2 queue[QCAP];
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

// base length and extra-bit count of length symbol 257+i (deflate.lisp:14-23)
__device__ __forceinline__ void length_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    if (i < 8) { extra = 0; base = 3 + i; }
    else if (i == 28) { extra = 0; base = 258; }
    else { extra = (i >> 2) - 1; base = 3 + ((4 + (i & 3)) << extra); }
}

__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t len)
{
    if (sym < 256) return (sym << 8) | LIT_FLAG | len;
    if (sym == 256) return (7u << 5) | len;
    if (sym > 285) return ((sym - 257) << 8) | (7u << 5) | len;
    uint32_t base, extra;
    length_code(sym - 257, base, extra);
    return ((base - 3) << 8) | (extra << 5) | len;
}
// distance code c (deflate.lisp:25-34): c < 4: distance c + 1; else extra = c/2 - 1, base = 1 + ((2 | c&1) << extra)
__device__ __forceinline__ uint32_t dist_entry(uint32_t code, uint32_t len)
{
    if (code > 29) return (code << 11) | DIST_BAD | len;
    const uint32_t extra = code < 4 ? 0 : (code >> 1) - 1, hi = code < 4 ? code : 2 | (code & 1);
    return (code << 11) | (hi << 8) | (extra << 4) | len;
}
__device__ __forceinline__ uint32_t clen_entry(uint32_t sym, uint32_t len) { return (sym << 3) | len; }

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

// Group-parallel construction of one canonical tree (lengths->ht,
// huffman.lisp:142-225): per-length counts (one lane per length), the
// exact-completeness check (huffman.lisp:166-184), the sorted symbol list and
// the root table.
template <int ROOT_BITS, typename SortT, typename RootT, typename MakeEntry>
__device__ void build_tree(const uint8_t *lens, int n, TreeMeta &m, SortT *sorted, RootT *root,
                           MakeEntry make_entry)
{
    const int gl = Gr::gl();
    static_assert(G >= 16, "one lane per code length");
    // 1. counts
    uint32_t c = 0;
    if (gl >= 1)
        for (int s = 0; s < n; s++) c += (lens[s] == gl);
    m.count[gl] = (uint16_t)c;
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
    if (gl >= 1) {
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

// LSB-first bit reader (lsb-bit-reader, bits.lisp:118), one per group leader,
// over the unit's LDS input ring.  Positions are "v-offsets": byte offset
// inside `in` plus the 16-byte misalignment of the `in` pointer, so that
// v % 16 == address % 16 and the ring is filled with aligned 16-byte loads
// (stage_* below).  The ring holds zero bytes past the end of the unit's input
// (bits.lisp:49-50); near_end turns on the exact end-of-input accounting that
// decides `eof` (bits.lisp:35-45).
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
    __device__ __forceinline__ uint32_t take()
    {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(ring + ((uint32_t)wpos & (IN_RING - 1)));
        if (wpos + 4 > limit) near_end = true;
        wpos += 4;
        return w;
    }
    // after refill(): cnt >= 33
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            if (skip) {
                buf = (uint64_t)take() >> skip;
                cnt = 32 - skip;
                skip = 0;
                if (cnt > 32) return;
            }
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

// ---- input staging: HBM -> LDS ring, one 16-byte block per lane (256 bytes per group step) ----
// The load is unconditional (address clamped into the unit's input) so that its result is not
// touched before stage_store(); the wait for it then falls after the work placed in between.
__device__ __forceinline__ uint4 stage_load(const uint8_t *base16, uint64_t v, uint64_t limit)
{
    const uint64_t last = limit ? (limit - 1) & ~15ull : 0;
    return *reinterpret_cast<const uint4 *>(base16 + (v < last ? v : last));
}
__device__ __forceinline__ void stage_store(uint8_t *inring, uint4 x, uint64_t v, uint64_t limit)
{
    if (v + 16 > limit) {                               // zero the bytes that do not exist
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t a = v + 4 * k;
            w[k] = a >= limit ? 0u : (a + 4 > limit ? w[k] & ((1u << (8 * (uint32_t)(limit - a))) - 1u) : w[k]);
        }
        x = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4 *>(inring + ((uint32_t)v & (IN_RING - 1))) = x;
}
// where staging continues for a reader whose next unread bit lies in the dword at wpos: restart at the reader if it jumped (stored blocks, headers)
__device__ __forceinline__ void stage_rebase(uint64_t wpos, uint64_t &s_hi)
{
    const uint64_t wfloor = wpos & ~15ull;
    if (s_hi < wfloor || s_hi > wfloor + IN_RING) s_hi = wfloor;
}
// fills the ring completely ahead of wpos; the load latency is exposed (unit start, table headers)
__device__ __forceinline__ void stage_blocking(uint8_t *inring, const uint8_t *base16, uint64_t limit, uint64_t wpos,
                                               uint64_t &s_hi)
{
    const int gl = Gr::gl();
    stage_rebase(wpos, s_hi);
    const uint64_t wfloor = wpos & ~15ull;
    while (s_hi + STAGE_BYTES <= wfloor + IN_RING) {
        const uint64_t v = s_hi + 16u * gl;
        stage_store(inring, stage_load(base16, v, limit), v, limit);
        s_hi += STAGE_BYTES;
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

__device__ __forceinline__ uint32_t ld_bytes_le(const uint8_t *p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// one byte of history at output position s (s < pos0): LDS ring, preset dictionary or HBM
__device__ __forceinline__ uint32_t history_byte(int64_t s, int64_t near_lo, const uint8_t *ring, uint32_t shift,
                                                 uint8_t *ob, const uint8_t *prefix_end)
{
    if (s >= near_lo) return ring[(uint32_t)((uint64_t)s + shift) & RING_MASK];
    if (s < 0) return prefix_end[s];
    // written by this wave earlier; read it from L2, not from a possibly stale L1 line
    return __hip_atomic_load(ob + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct bool_false { static constexpr bool value = false; };
struct bool_true { static constexpr bool value = true; };

__global__ __launch_bounds__(64, 1) void inflate_kernel(const uint8_t *in, const semz_unit *units,
                                                        uint32_t nunits, uint8_t *out,
                                                        semz_result *results, semz_tail *tails)
{
    __shared__ UnitLds lds_all[UPW];
    const int gl = Gr::gl();
    const bool leader = gl == 0;
    const uint32_t u = blockIdx.x * UPW + (Gr::lane() / G);
    const bool have_unit = u < nunits;
    UnitLds &L = lds_all[Gr::lane() / G];

    semz_unit un;
    if (have_unit) un = units[u];
    else un = semz_unit{};

    uint8_t *ob = out + un.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);      // ring index == global address (mod 16)
    const uint64_t out_cap = un.out_cap;

    // ---- leader-only state ----
    // v-offsets (see BitReader): offset inside `in` + 16-byte misalignment of the `in` pointer
    const uint64_t amis = (uint64_t)((uintptr_t)in & 15);
    const uint8_t *base16 = in - amis;              // base16 + v is the address of v-offset v
    uint64_t s_hi = 0;                               // input staged up to this v-offset (group-uniform)
    uint8_t *safe_byte = const_cast<uint8_t *>(base16);  // always-readable address for masked-off gather lanes
    const uint64_t limit_v = un.in_limit + amis;
    BitReader br;
    Status st{SEMZ_OK, 0};
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

    uint64_t pos = 0;                    // output bytes produced (all lanes track it)
    uint64_t flushed = 0;                // output bytes already in HBM
    bool alive = have_unit;

    // ---- container header (leader) ----
    br.init(L.inring, limit_v, limit_v);
    if (leader && alive) {
        br.init(L.inring, un.in_off + amis, limit_v);
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
                            aux = __builtin_bswap32(ld_bytes_le(in + p, 4));
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
    prefix_end = (const uint8_t *)(uintptr_t)Gr::bcast((uint64_t)(uintptr_t)prefix_end);
    if (alive) stage_blocking(L.inring, base16, limit_v, Gr::bcast((br.bitpos() >> 5) << 2), s_hi);

    STAMP_DECL
    // ---- main loop: one round per iteration ----
    while (__any(alive)) {
        int ev = EV_DONE;
        uint32_t ntok = 0, rbytes = 0;
        STAMP(0);
        if (leader && alive) {
            ev = st.code != SEMZ_OK ? EV_DONE : EV_ROUND;
            const uint32_t room = (uint32_t)(out_cap - pos < 0xFFFFFFFFull ? out_cap - pos : 0xFFFFFFFFull);
            const uint32_t avail = (uint32_t)(prefix_amount + pos < 0x7FFFFFFFull ? prefix_amount + pos : 0x7FFFFFFFull);
            const uint32_t rpos = (uint32_t)pos + shift;
#define FAIL(c, d) { st.set((c), (d)); ev = EV_DONE; return; }
            // one literal/length symbol (+ distance), decode-huffman-data deflate.lisp:313-338;
            // CHK carries the end-of-input checks, which only matter once the reader is near the end
            auto huff_step = [&](auto CHK) {
                constexpr bool chk = decltype(CHK)::value;
                uint32_t e = L.lit_root[br.peek(LIT_ROOT_BITS)];
                uint32_t cl = e & 15;
                if (cl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    cl = canon_find(L.lit, v15, LIT_ROOT_BITS + 1, 15, idx);
                    e = lit_entry(L.lit_sorted[idx], cl);
                }
                if constexpr (chk) if (br.short_of(lit_full ? lit_max : (int)cl)) FAIL(SEMZ_EOF, 0)
                br.drop(cl);
                if (e & LIT_FLAG) {
                    if (rbytes >= room) FAIL(SEMZ_E_OUT_CAP, 0)
                    L.ring[(rpos + rbytes) & RING_MASK] = (uint8_t)(e >> 8);
                    rbytes++;
                    return;
                }
                const uint32_t lx = (e >> 5) & 7;
                if (lx == 7) {
                    if ((e >> 8) == 0) { state = ST_HDR; return; }
                    FAIL(SEMZ_E_DEFLATE_LITLEN_CODE, 257 + (e >> 8))
                }
                if (dist_mode == MODE_LITONLY) {
                    uint32_t sym = 257;
                    for (uint32_t i = 0; i < 29; i++) { uint32_t b, x; length_code(i, b, x); if (b == 3 + (e >> 8)) sym = 257 + i; }
                    FAIL(SEMZ_E_DEFLATE_LEN_IN_LITONLY, sym)
                }
                if constexpr (chk) if (br.short_of(lx)) FAIL(SEMZ_EOF, 0)
                const uint32_t length = 3 + (e >> 8) + br.peek(lx);
                br.drop(lx);
                br.refill();
                uint32_t d = L.dist_root[br.peek(DIST_ROOT_BITS)];
                uint32_t dl = d & 15;
                if (dl == 0) {
                    uint32_t v15 = __brev((uint32_t)br.buf) >> 17, idx = 0;
                    dl = canon_find(L.dist, v15, DIST_ROOT_BITS + 1, 15, idx);
                    d = dist_entry(L.dist_sorted[idx], dl);
                }
                if constexpr (chk) if (br.short_of(dist_full ? dist_max : (int)dl)) FAIL(SEMZ_EOF, 0)
                br.drop(dl);
                if (d & DIST_BAD) FAIL(dist_mode == MODE_SINGLE ? SEMZ_E_DEFLATE_SINGLE_DIST : SEMZ_E_DEFLATE_DIST_CODE, d >> 11)
                const uint32_t dx = (d >> 4) & 15;
                if constexpr (chk) if (br.short_of(dx)) FAIL(SEMZ_EOF, 0)
                const uint32_t distance = 1 + (((d >> 8) & 3) << dx) + br.peek(dx);
                br.drop(dx);
                const uint32_t have = avail + rbytes;                           // deflate.lisp:329-331
                if (distance > (have < wsize ? have : wsize)) FAIL(SEMZ_E_DEFLATE_DISTANCE, distance)
                if (rbytes + length > room) FAIL(SEMZ_E_OUT_CAP, 0)
                L.queue[ntok] = make_uint2(T_MATCH | (length << 2) | (rbytes << 16), distance);
                ntok++;
                rbytes += length;
            };
#undef FAIL
#define FAIL(c, d) { st.set((c), (d)); ev = EV_DONE; break; }
            // every iteration takes at most eight dwords from the input ring
            while (ev == EV_ROUND && ntok < (uint32_t)QCAP && rbytes < (uint32_t)ROUND_BYTES && br.wpos + 32 <= s_hi) {
                if (state == ST_HUFF && !br.near_end && dist_mode != MODE_LITONLY) {
                    // ---- fast path: tokens that need no end-of-input accounting, no long codes and no
                    // block events.  The bit window is two dwords of the input ring read with
                    // v_alignbit; anything unusual leaves the loop before the token is committed and
                    // is decoded again by the general code below.
                    const uint64_t bits0 = br.bitpos();
                    const uint64_t wlo0 = (bits0 >> 5) << 2;                   // v-offset of the dword holding the next bit
                    const uint64_t e_end = s_hi < (limit_v & ~3ull) ? s_hi : (limit_v & ~3ull);
                    const uint32_t end32 = (uint32_t)e_end, wp0 = (uint32_t)wlo0;
                    uint32_t wp = wp0, bo = (uint32_t)bits0 & 31;
                    if (e_end >= wlo0 + 16) {
                        uint32_t lo = *reinterpret_cast<const uint32_t *>(L.inring + (wp & (IN_RING - 1)));
                        uint32_t hi = *reinterpret_cast<const uint32_t *>(L.inring + ((wp + 4) & (IN_RING - 1)));
                        const uint32_t rb_lim = room >= 258u + ROUND_BYTES ? (uint32_t)ROUND_BYTES : (room >= 258u ? (room - 258u < (uint32_t)ROUND_BYTES ? room - 258u : (uint32_t)ROUND_BYTES) : 0u);
                        while ((int32_t)(end32 - wp) >= 16 && rbytes < rb_lim && ntok < (uint32_t)QCAP) {
                            if (bo >= 32) {
                                lo = hi;
                                hi = *reinterpret_cast<const uint32_t *>(L.inring + ((wp + 8) & (IN_RING - 1)));
                                wp += 4;
                                bo -= 32;
                            }
                            const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, bo);
                            const uint32_t e = L.lit_root[x & ((1u << LIT_ROOT_BITS) - 1)];
                            const uint32_t cl = e & 15;
                            if (cl == 0) break;
                            if (e & LIT_FLAG) {
                                L.ring[(rpos + rbytes) & RING_MASK] = (uint8_t)(e >> 8);
                                rbytes++;
                                bo += cl;
                                continue;
                            }
                            const uint32_t lx = (e >> 5) & 7;
                            if (lx == 7) break;
                            const uint32_t length = 3 + (e >> 8) + __builtin_amdgcn_ubfe(x, cl, lx);
                            uint32_t bo2 = bo + cl + lx, lo2 = lo, hi2 = hi, wp2 = wp;
                            if (bo2 >= 32) {
                                lo2 = hi;
                                hi2 = *reinterpret_cast<const uint32_t *>(L.inring + ((wp + 8) & (IN_RING - 1)));
                                wp2 += 4;
                                bo2 -= 32;
                            }
                            const uint32_t y = __builtin_amdgcn_alignbit(hi2, lo2, bo2);
                            const uint32_t d = L.dist_root[y & ((1u << DIST_ROOT_BITS) - 1)];
                            const uint32_t dl = d & 15;
                            if (dl == 0 || (d & DIST_BAD)) break;
                            const uint32_t dx = (d >> 4) & 15;
                            const uint32_t distance = 1 + (((d >> 8) & 3) << dx) + __builtin_amdgcn_ubfe(y, dl, dx);
                            const uint32_t have = avail + rbytes;
                            if (distance > (have < wsize ? have : wsize)) break;
                            lo = lo2; hi = hi2; wp = wp2;
                            bo = bo2 + dl + dx;
                            L.queue[ntok] = make_uint2(T_MATCH | (length << 2) | (rbytes << 16), distance);
                            ntok++;
                            rbytes += length;
                        }
                        // hand the position back to the general reader
                        const uint64_t bits1 = 8 * (wlo0 + (uint64_t)(wp - wp0)) + bo;
                        br.wpos = (bits1 >> 5) << 2;
                        br.skip = (int)(bits1 & 31);
                        br.buf = 0;
                        br.cnt = 0;
                        if (!(ntok < (uint32_t)QCAP && rbytes < (uint32_t)ROUND_BYTES && br.wpos + 32 <= s_hi)) break;
                    }
                }
                br.refill();
                if (state == ST_HUFF) {
                    if (!br.near_end) huff_step(bool_false{});
                    else huff_step(bool_true{});
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
                            if ((uint64_t)rbytes + len > room) FAIL(SEMZ_E_OUT_CAP, 0)
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
                        for (int i = 0; i < 19; i++) L.clens[i] = 0;
                        bool eof = false;
                        for (int i = 0; i < nclen; i++) {
                            // code-length order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (deflate.lisp:174)
                            const uint64_t ORDER_MID = 0x0B050A0609070800ull;   // entries 3..10
                            uint32_t sym;
                            if (i < 3) sym = 16 + i;
                            else if (i < 11) sym = (uint32_t)(ORDER_MID >> (8 * (i - 3))) & 0xFF;
                            else sym = (i & 1) ? (uint32_t)(19 - i) / 2u : (uint32_t)(i + 12) / 2u;
                            br.refill();
                            if (br.short_of(3)) { eof = true; break; }
                            L.clens[sym] = (uint8_t)br.peek(3);
                            br.drop(3);
                        }
                        if (eof) FAIL(SEMZ_EOF, 0)
                        ev = EV_DYN;
                        break;
                    } else FAIL(SEMZ_E_DEFLATE_BTYPE, 0)
                    continue;
                }
                // ---- ST_STORED: raw bytes; the bit buffer is byte aligned, hand the span to the group ----
                {
                    const uint32_t room_r = ROUND_BYTES - rbytes;
                    const uint32_t n = stored_left < room_r ? stored_left : room_r;
                    const uint64_t src = br.bitpos() >> 3;            // v-offset of the span
                    L.queue[ntok] = make_uint2(T_RAW | (n << 2) | (rbytes << 16), (uint32_t)(src - amis - un.in_off));
                    ntok++;
                    rbytes += n;
                    stored_left -= n;
                    br.init(L.inring, src + n, limit_v);                 // restart the bit buffer after the span
                    if (stored_left == 0) state = ST_HDR;
                }
            }
#undef FAIL
        }
        STAMP(1);
        ev = Gr::bcast(ev);
        ntok = Gr::bcast(ntok);
        rbytes = Gr::bcast(rbytes);
        lds_sync();
        COUNT(0, leader ? ntok : 0); COUNT(1, 1);

        // ---- top up the input ring: the loads are issued here and stored after the resolve ----
        const uint64_t wp = Gr::bcast((br.bitpos() >> 5) << 2);
        stage_rebase(wp, s_hi);
        const uint32_t nst = (uint32_t)(((wp & ~15ull) + IN_RING - s_hi) / STAGE_BYTES);
        const uint64_t sv0 = s_hi + 16u * gl, sv1 = sv0 + STAGE_BYTES;
        const uint4 sx0 = stage_load(base16, sv0, limit_v), sx1 = stage_load(base16, sv1, limit_v);

        // ---- resolve the round's tokens ----
        const uint64_t pos0 = pos, pos1 = pos + rbytes;
        // positions >= near_lo live in the ring; older ones are in HBM (flushed), negative ones in the preset dictionary
        const int64_t near_lo = (int64_t)pos1 > RING ? (int64_t)pos1 - RING : 0;
        if (alive && ntok) {
            uint32_t farm = 0, nearm = 0, rawm = 0;
#pragma unroll
            for (int h = 0; h < QCAP / G; h++) {
                const uint32_t t = (uint32_t)gl + G * h;
                const uint2 tk = L.queue[t];
                const bool valid = t < ntok;
                const bool is_match = valid && (tk.x & 3) == T_MATCH;
                const bool far = is_match && ((int64_t)(pos0 + (tk.x >> 16)) - (int64_t)tk.y) < near_lo;
                farm |= (uint32_t)Gr::ballot(far) << (G * h);
                nearm |= (uint32_t)Gr::ballot(is_match && !far) << (G * h);
                rawm |= (uint32_t)Gr::ballot(valid && (tk.x & 3) == T_RAW) << (G * h);
            }
            COUNT(2, leader ? __popc(farm) : 0); COUNT(3, leader ? __popc(nearm) : 0);
            STAMP(2);
            // far matches: no byte of the source was produced in this round, so the gathers of four
            // tokens (32 bytes each) are put in flight together
            while (farm) {
                uint32_t fl[4], fp[4], gv[4][2], lv[4][2];
                int64_t fs[4];
                bool neg = false;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t more = farm;
                    const uint2 tk = L.queue[more ? __builtin_ctz(more) : 0];
                    farm &= farm - 1;
                    fl[b] = more ? (tk.x >> 2) & 0x3FFF : 0;
                    fp[b] = tk.x >> 16;
                    fs[b] = (int64_t)(pos0 + fp[b]) - (int64_t)tk.y;
                }
                // unconditional loads (HBM through L2: the bytes were written by this wave; LDS ring), selected below
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int64_t s = fs[b] + gl + G * h;
                        const bool act = (uint32_t)gl + G * h < fl[b];
                        const bool useg = act && s >= 0 && s < near_lo;
                        gv[b][h] = __hip_atomic_load(useg ? ob + s : safe_byte, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lv[b][h] = L.ring[(uint32_t)((uint64_t)s + shift) & RING_MASK];
                        neg = neg || (act && s < 0);
                    }
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int64_t s = fs[b] + gl + G * h;
                        if ((uint32_t)gl + G * h < fl[b] && s >= 0)
                            L.ring[(uint32_t)(pos0 + fp[b] + gl + G * h + shift) & RING_MASK] =
                                (uint8_t)(s >= near_lo ? lv[b][h] : gv[b][h]);
                    }
                if (__any(neg)) {                      // preset dictionary bytes (zlib FDICT, deflate prefix)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        for (uint32_t k = gl; k < fl[b] && k < 2 * G; k += G)
                            if (fs[b] + k < 0) L.ring[(uint32_t)(pos0 + fp[b] + k + shift) & RING_MASK] = prefix_end[fs[b] + k];
                }
#pragma unroll
                for (int b = 0; b < 4; b++)
                    for (uint32_t k = 2 * G + gl; k < fl[b]; k += G)
                        L.ring[(uint32_t)(pos0 + fp[b] + k + shift) & RING_MASK] =
                            (uint8_t)history_byte(fs[b] + k, near_lo, L.ring, shift, ob, prefix_end);
            }
            STAMP(3);
            // stored spans straight from the input
            while (rawm) {
                const int j = __builtin_ctz(rawm);
                rawm &= rawm - 1;
                const uint2 tk = L.queue[j];
                const uint32_t len = (tk.x >> 2) & 0x3FFF;
                const uint32_t p = (uint32_t)(pos0 + (tk.x >> 16)) + shift;
                const uint8_t *s = in + un.in_off + tk.y;
                for (uint32_t k = gl; k < len; k += G) L.ring[(p + k) & RING_MASK] = s[k];
            }
            // near matches in token order; LDS operations of one wave execute in order
            if (nearm) {
                uint2 nx = L.queue[__builtin_ctz(nearm)];
                while (nearm) {
                    const uint2 tk = nx;
                    nearm &= nearm - 1;
                    if (nearm) nx = L.queue[__builtin_ctz(nearm)];
                    const uint32_t dist = tk.y;
                    const uint32_t len = (tk.x >> 2) & 0x3FFF;
                    const uint32_t p = (uint32_t)(pos0 + (tk.x >> 16)) + shift;     // ring index of the destination
                    const uint32_t src = p - dist;
                    if (dist >= len) {
                        for (uint32_t k = gl; k < len; k += G)
                            L.ring[(p + k) & RING_MASK] = L.ring[(src + k) & RING_MASK];
                    } else if (dist == 1) {
                        const uint8_t b = L.ring[src & RING_MASK];
                        for (uint32_t k = gl; k < len; k += G) L.ring[(p + k) & RING_MASK] = b;
                    } else {
                        for (uint32_t k = gl; k < len; k += G)
                            L.ring[(p + k) & RING_MASK] = L.ring[(src + k % dist) & RING_MASK];
                    }
                }
            }
        }
        pos = pos1;
        lds_sync();
        STAMP(4);

        // ---- flush the ring ----
        if (alive) {
            const bool last = ev == EV_DONE;
            const uint32_t head = (16 - shift) & 15;
            if (flushed == 0 && head && (pos >= head || last)) {
                const uint32_t n = pos < head ? (uint32_t)pos : head;
                if ((uint32_t)gl < n) ob[gl] = L.ring[(gl + shift) & RING_MASK];
                flushed = n;
            }
            if (flushed || !head) {
                while (pos - flushed >= (uint64_t)FLUSH_BYTES) {
                    const uint64_t a = flushed + 16u * gl;
                    const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                    *reinterpret_cast<uint4 *>(ob + a) = v;
                    flushed += FLUSH_BYTES;
                }
                if (last) {
                    const uint32_t pieces = (uint32_t)((pos - flushed) >> 4);
                    if ((uint32_t)gl < pieces) {
                        const uint64_t a = flushed + 16u * gl;
                        const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & RING_MASK]);
                        *reinterpret_cast<uint4 *>(ob + a) = v;
                    }
                    flushed += 16ull * pieces;
                    const uint32_t rest = (uint32_t)(pos - flushed);
                    if ((uint32_t)gl < rest) ob[flushed + gl] = L.ring[(uint32_t)(flushed + gl + shift) & RING_MASK];
                    flushed = pos;
                }
            }
        }
        if (alive) {
            if (nst >= 1) { stage_store(L.inring, sx0, sv0, limit_v); s_hi += STAGE_BYTES; }
            if (nst >= 2) { stage_store(L.inring, sx1, sv1, limit_v); s_hi += STAGE_BYTES; }
        }
        lds_sync();
        STAMP(5);

        // ---- table construction for the next block ----
        if (alive && (ev == EV_DYN || ev == EV_FIXED)) {
            long overread = 0;
            if (ev == EV_DYN) {
                // code-length tree, then the code lengths themselves (deflate.lisp:176-197)
                build_tree<CLEN_ROOT_BITS>(L.clens, 19, L.clen, L.clen_sorted, L.clen_root, clen_entry);
                const int tstatus = (int)L.clen.status;
                // the code lengths take at most 320 * 14 bits: make sure they are staged
                stage_blocking(L.inring, base16, limit_v, Gr::bcast((br.bitpos() >> 5) << 2), s_hi);
                if (leader) {
                    overread = (final_block ? 0 : MIN_BLOCK_BITS) + trailing_bits;    // deflate.lisp:392-396
                    if (tstatus == SEMZ_OK) {
                        const bool cfull = overread >= (long)(L.clen.maxlen - L.clen.minlen);
                        const int cmax = (int)L.clen.maxlen, total = nlit + ndist;
                        int i = 0;
                        while (i < total) {
                            br.refill();
                            const uint32_t ce = L.clen_root[br.peek(CLEN_ROOT_BITS)];
                            const int cl = ce & 7;
                            if (br.short_of(cfull ? cmax : cl)) { st.set(SEMZ_EOF, 0); break; }
                            br.drop(cl);
                            const uint32_t sym = ce >> 3;
                            if (sym < 16) { L.lens[i++] = (uint8_t)sym; continue; }
                            const int xb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                            if (br.short_of(xb)) { st.set(SEMZ_EOF, 0); break; }
                            const int n = (sym == 18 ? 11 : 3) + (int)br.peek(xb);
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
            const int ok = Gr::bcast((int)(st.code == SEMZ_OK));
            nlit = Gr::bcast(nlit);
            ndist = Gr::bcast(ndist);
            lds_sync();
            if (ok) {
                build_tree<LIT_ROOT_BITS>(L.lens, nlit, L.lit, L.lit_sorted, L.lit_root, lit_entry);
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
                        lds_sync();
                        for (int i = gl; i < 32; i += G) {
                            uint8_t v = 0;
                            if ((mode & 255) == MODE_LITONLY) v = i >= 30;
                            else { const int one = mode >> 8; v = i == one || i == (one == 31 ? 30 : 31); }
                            L.lens[i] = v;
                        }
                        lds_sync();
                        build_tree<DIST_ROOT_BITS>(L.lens, 32, L.dist, L.dist_sorted, L.dist_root, dist_entry);
                    } else {
                        build_tree<DIST_ROOT_BITS>(dl, ndist, L.dist, L.dist_sorted, L.dist_root, dist_entry);
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
        STAMP(6);

        // ---- unit end: trailer and results (leader) ----
        if (ev == EV_DONE && alive) {
            if (leader) {
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
#ifdef SEMZ_STAMPS
                for (int i = 0; i < 8; i++) atomicAdd(&g_stamps[i], (unsigned long long)t_acc[i]);
                for (int i = 0; i < 4; i++) atomicAdd(&g_stamps[8 + i], (unsigned long long)n_acc[i]);
                atomicAdd(&g_stamps[12], 1ull);
#endif
            }
            alive = false;
        }
    }
}

} // namespace

#ifdef SEMZ_STAMPS
extern "C" int semz_debug_stamps(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" void semz_launch_inflate(const uint8_t *d_in, const semz_unit *d_units, uint32_t nunits,
                                    uint8_t *d_out, semz_result *d_results, semz_tail *d_tails,
                                    hipStream_t stream)
{
    if (nunits == 0) return;
    inflate_kernel<<<(nunits + UPW - 1) / UPW, 64, 0, stream>>>(d_in, d_units, nunits, d_out, d_results, d_tails);
}
