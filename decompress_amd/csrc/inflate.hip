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
// Layout: one unit per 64-lane wavefront, one wavefront per workgroup, about
// 10 KiB of LDS per unit, so a CU holds 16 units (four waves per SIMD).  All
// control state is wave-uniform (SGPRs, scalar branches); the lanes work on the
// data-parallel parts.  A lone wave issues one instruction every ~8 clocks
// whatever its kind and four waves of a SIMD together get ~2.7 times that, so
// the design goal is few instructions per token:
//
// Window decode.  Lane l decodes the complete token that would start at bit
// P + l of the input: lit/len root lookup, length extra bits, distance root
// lookup, distance extra bits -- at most 48 bits, all inside the lane's own
// 64-bit window of the LDS input ring.  The wave then follows the chain
// off -> off + token_bits(off) with readlane and collects the token starts of
// the next 64 bits in a mask.  Output offsets are a DPP prefix sum over the
// start lanes; literals go straight into the unit's LDS output ring, matches
// into an LDS queue (rank = mbcnt of the match mask).  Anything unusual (code
// longer than the root index, end of block, invalid code, distance too far,
// output full, the last bytes of the input) cuts the window at that token,
// which is then decoded by the scalar step with the reference's checks.
//
// Rounds.  A round ends after 2 KiB of output or 64 queued matches; the wave
// then resolves the queue:
//   far matches     (source older than the ring, or in the preset dictionary)
//                   gathered from HBM, four tokens x 64 bytes in flight
//   stored spans    copied from the input
//   near matches    in token order, one wave-wide LDS copy each (periodic
//                   replication when distance < length, common.lisp:324-333);
//                   LDS operations of a wave execute in order, so consecutive
//                   copies need no barrier
// The ring index equals the HBM address modulo 16 and the ring is flushed with
// 16-byte stores.  Huffman tables are rebuilt in LDS per block: a root lookup
// table plus a canonical (first-code / sorted-symbol) description used for
// codes longer than the root and for the table fill.
#include "semz_device.h"
#include <type_traits>

#ifdef SEMZ_STAMPS
// diagnostic build only (make STAMPS=1): per-phase shader-clock totals summed over all units
__device__ unsigned long long g_stamps[16];
// per unit (the first 16384 of a launch): start and end on the constant 100 MHz clock, and where the wave ran (HW_ID)
__device__ unsigned long long g_unit_times[16384][3];
#define STAMP_DECL const uint64_t t_unit0 = __builtin_amdgcn_s_memrealtime(); uint64_t t_last = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0}; uint32_t n_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { uint64_t t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define COUNT(i, n) do { n_acc[i] += (n); } while (0)
#else
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define COUNT(i, n) do { } while (0)
#endif

namespace {

constexpr int LIT_ROOT_BITS = 10;
constexpr int DIST_ROOT_BITS = 8;
constexpr int CLEN_ROOT_BITS = 7;
constexpr int PRIO_TURN_SHIFT = 14;      // a turn at a priority lasts 2^14 ticks of the 100 MHz clock, 164 us (asked every round: 5 us 9.35 ms; 20 us 8.86; 164 and 655 us 8.81; 2.6 ms 9.01)
constexpr int RING = 4096;               // output staging ring (bytes)
constexpr int RING_MASK = RING - 1;
constexpr int ROUND_BYTES = 2048;        // a round stops once it has produced this many bytes
constexpr int FLUSH_BYTES = 16 * 64;     // one 16-byte store per lane
constexpr int QCAP = 128;                // match / stored tokens per round (two per lane when they are resolved)
constexpr int IN_RING = 1024;            // compressed-input staging ring (bytes)
constexpr int IN_PIECE = 512;            // staged in pieces of 64 lanes x 8 bytes
constexpr int TOKEN_BITS = 48;           // longest token: 15 + 5 + 15 + 13
constexpr int MIN_BLOCK_BITS = 10;       // +min-deflate-block-bitsize+, deflate.lisp:4-12
static_assert(RING - ROUND_BYTES - 258 >= FLUSH_BYTES + 258, "far sources must already be flushed");

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
    uint8_t inring[IN_RING + 16];        // the first 16 bytes again behind the end: a window never wraps
    uint2 queue[QCAP];
    uint16_t lit_root[1 << LIT_ROOT_BITS];
    uint16_t dist_root[1 << DIST_ROOT_BITS];
    uint16_t lit_sorted[288];
    uint8_t dist_sorted[32];
    uint8_t clen_root[1 << CLEN_ROOT_BITS];
    uint8_t lens[320];                   // (the code-length tree's own description lives in here while it is needed:
    uint8_t clens[32];                   //  CLEN_SORTED_AT, CLEN_META_AT -- it is dead before the first length is stored)
    uint8_t flags[64];
    TreeMeta lit, dist;
};
constexpr int CLEN_SORTED_AT = 96, CLEN_META_AT = 128;   // offsets into UnitLds::lens
static_assert(CLEN_SORTED_AT + 32 <= CLEN_META_AT && CLEN_META_AT % 8 == 0 && CLEN_META_AT + sizeof(TreeMeta) <= 320, "the code-length tree fits `lens`");
static_assert(sizeof(UnitLds) <= 10240, "sixteen units per CU");
static_assert(__builtin_offsetof(UnitLds, lens) % 8 == 0 && sizeof(((UnitLds *)nullptr)->lens) >= 8 * 40 &&
              8 * 40 * 8 >= ROUND_BYTES + 258, "the resolve step's bitmap of match starts lives in `lens`");
static_assert(__builtin_offsetof(UnitLds, clen_root) % 8 == 0 &&
              __builtin_offsetof(UnitLds, lit) - __builtin_offsetof(UnitLds, clen_root) >= 512, "64 dummy queue slots fit the table scratch");

typedef const __attribute__((address_space(4))) uint32_t *k_u32p;   // scalar-load view of read-only memory

// the lane's bit of a wave-uniform mask as a lane predicate (the mask becomes the condition operand itself)
__device__ __forceinline__ bool lane_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

template <typename T> __device__ __forceinline__ T uni(T v)           // says: this value is the same in every lane
{
    if constexpr (sizeof(T) == 8) {
        uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
        return (T)(((uint64_t)hi << 32) | lo);
    } else {
        return (T)__builtin_amdgcn_readfirstlane((int)v);
    }
}

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
// Root tables hold 0 ("ask the scalar step") for everything the window decode does not take itself: end of
// block, 286/287 and distance codes 30/31, besides the codes longer than the root index.
__device__ __forceinline__ uint32_t lit_root_entry(uint32_t sym, uint32_t len) { return sym == 256 || sym > 285 ? 0u : lit_entry(sym, len); }
__device__ __forceinline__ uint32_t dist_root_entry(uint32_t code, uint32_t len) { return code > 29 ? 0u : dist_entry(code, len); }

// Canonical decode of a left-aligned 15-bit MSB-first window, lengths lo..hi.
// Returns the code length (0 if none in range) and the symbol's index in sorted[].
__device__ __forceinline__ uint32_t canon_find(const TreeMeta &m, uint32_t v15, int lo, int hi, uint32_t &idx)
{
    for (int l = lo; l <= hi; l++) {
        if (v15 < uni(m.limit[l])) {
            idx = uni((uint32_t)m.offs[l]) + (v15 >> (15 - l)) - uni((uint32_t)m.first[l]);
            return (uint32_t)l;
        }
    }
    return 0;
}

// Placed right behind a lane-masked store: the two paths of the mask then join here, in a block without phis.
// Without it the optimiser lets each path jump on its own to the next merge of scalar state (a loop latch,
// say), the uniformity analysis sees scalars merged behind a divergent branch, and the decoder state moves
// from SGPRs to VGPRs.  Emits no instruction.
__device__ __forceinline__ void join_lanes() { __builtin_amdgcn_wave_barrier(); }

// Wave-parallel construction of one canonical tree (lengths->ht,
// huffman.lisp:142-225): per-length counts (one lane per length), the
// exact-completeness check (huffman.lisp:166-184), the sorted symbol list and
// the root table.
template <int ROOT_BITS, typename SortT, typename RootT, typename MakeEntry>
__device__ __forceinline__ void build_tree(const uint8_t *lens, int n, TreeMeta &m, SortT *sorted, RootT *root,
                                           MakeEntry make_entry)
{
    const int lane = threadIdx.x & 63;
    // 1. counts, lanes = symbols: four ballots find the lanes of a piece that hold the same length, and the first
    //    of each such set adds the set's size to the length's counter (one writer per counter: no atomics)
    const uint64_t below = (1ull << lane) - 1;
    if (lane < 16) m.count[lane] = 0;
    lds_sync();
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t len = s < n ? lens[s] : 0u;
        uint64_t same = __builtin_amdgcn_ballot_w64(len != 0);
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const uint64_t bm = __builtin_amdgcn_ballot_w64((len >> bit) & 1);
            same &= ((len >> bit) & 1) ? bm : ~bm;
        }
        if (len && (same & below) == 0) m.count[len] = (uint16_t)(m.count[len] + (uint32_t)__builtin_popcountll(same));
        join_lanes();
        lds_sync();
    }
    // 2. canonical bookkeeping, lanes = lengths: limit[l] = sum over k <= l of count[k] << (15 - k) is a prefix sum,
    //    first[l] = limit[l - 1] >> (15 - l), offs[l] the counts in front (one lane walking the fifteen lengths took
    //    a quarter of the three trees' time)
    {
        const uint32_t cnt = lane >= 1 && lane <= 15 ? m.count[lane & 15] : 0u;
        const uint32_t wgt = cnt << (15 - (lane & 15));
        uint32_t iw = wgt, ic = cnt;
#define ROW_SCAN(v, ctl) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctl, 0xf, 0xf, false)
        ROW_SCAN(iw, 0x111); ROW_SCAN(ic, 0x111);      // row_shr:1, 2, 4, 8: the sixteen lanes of the first row
        ROW_SCAN(iw, 0x112); ROW_SCAN(ic, 0x112);
        ROW_SCAN(iw, 0x114); ROW_SCAN(ic, 0x114);
        ROW_SCAN(iw, 0x118); ROW_SCAN(ic, 0x118);
#undef ROW_SCAN
        if (lane < 16) {
            m.first[lane] = (uint16_t)(lane ? (iw - wgt) >> (15 - lane) : 0u);
            m.offs[lane] = (uint16_t)(ic - cnt);
            m.limit[lane] = iw;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)iw, 15);
        const uint32_t used = (uint32_t)__builtin_amdgcn_ballot_w64(cnt != 0);
        if (lane == 0) {
            m.minlen = used ? (uint32_t)__builtin_ctz(used) : 20u;
            m.maxlen = used ? 31u - (uint32_t)__builtin_clz(used) : 1u;
            m.status = total < 32768u ? SEMZ_E_HUFF_UNDERFULL : total > 32768u ? SEMZ_E_HUFF_OVERFULL : SEMZ_OK;
        }
        join_lanes();
    }
    lds_sync();
    if (uni(m.status) != SEMZ_OK) return;
    // 3. sorted symbol list, again lanes = symbols: a symbol's place is its length's running offset plus its rank
    //    among the lanes of the piece with the same length (m.count is the running offset from here on)
    if (lane < 16) m.count[lane] = m.offs[lane];
    lds_sync();
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t len = s < n ? lens[s] : 0u;
        uint64_t same = __builtin_amdgcn_ballot_w64(len != 0);
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const uint64_t bm = __builtin_amdgcn_ballot_w64((len >> bit) & 1);
            same &= ((len >> bit) & 1) ? bm : ~bm;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(same & below);
        const uint32_t at = len ? m.count[len] : 0u;
        join_lanes();
        lds_sync();
        if (len) {
            sorted[at + rank] = (SortT)s;
            if (rank == 0) m.count[len] = (uint16_t)(at + (uint32_t)__builtin_popcountll(same));
        }
        join_lanes();
        lds_sync();
    }
    // 4. root table, one index per lane per step.  The length of the code an index starts with is one more than
    //    the number of limits at or below it (the limits are in scalar registers, one compare and one add each; a
    //    search loop that left at the first limit above ran as long as the longest code for every lane), and what
    //    turns the code into its place in `sorted`, offs[l] - first[l], waits in `count`, free by now.
    if (lane < 16) m.count[lane] = (uint16_t)(m.offs[lane] - m.first[lane]);
    uint32_t lim[ROOT_BITS + 1];
#pragma unroll
    for (int l = 1; l <= ROOT_BITS; l++) lim[l] = uni(m.limit[l]);
    lds_sync();
    for (int i = lane; i < (1 << ROOT_BITS); i += 64) {
        const uint32_t v15 = (__brev((uint32_t)i) >> (32 - ROOT_BITS)) << (15 - ROOT_BITS);
        uint32_t l = 1;
#pragma unroll
        for (int k = 1; k < ROOT_BITS; k++) l += v15 >= lim[k];
        const uint32_t at = (uint16_t)(m.count[l] + (v15 >> (15 - l)));
        const RootT ent = (RootT)make_entry((uint32_t)sorted[at < (uint32_t)n ? at : 0u], l);
        root[i] = v15 < lim[ROOT_BITS] ? ent : (RootT)0;
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
__device__ __forceinline__ uint64_t parse_gzip_header(const uint8_t *in, uint64_t p0, uint64_t lim, Status &st)
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



// wave-wide inclusive prefix maximum of unsigned values (identity 0)
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// wave-wide inclusive prefix sum (DPP row shifts, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}

__global__ __launch_bounds__(64, 4) void inflate_kernel(const uint8_t *in, uint64_t in_len, const semz_unit *units,
                                                        uint32_t nunits, uint8_t *out, uint64_t out_total,
                                                        semz_result *results, semz_tail *tails)
{
    __shared__ UnitLds L;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const uint32_t lane = threadIdx.x & 63;
    semz_unit un;
    {
        k_u32p up = (k_u32p)(uintptr_t)(units + u);
        uint32_t w[sizeof(semz_unit) / 4];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(semz_unit) / 4); i++) w[i] = up[i];
        __builtin_memcpy(&un, w, sizeof un);
    }
    if (!semz_unit_spans_ok(un, in_len, out_total, 0)) {         // nothing of this unit is touched
        if (lane == 0) {
            semz_result r{};
            r.status = SEMZ_E_BAD_UNIT;
            r.out_off = un.out_off;
            results[u] = r;
            tails[u] = semz_tail{};
        }
        return;
    }

    uint8_t *ob = out + un.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);      // ring index == global address (mod 16)
    const uint64_t out_cap = un.out_cap;

    // v-offsets: offset inside `in` + 8-byte misalignment of the `in` pointer, so that the input ring is
    // filled with aligned 8-byte loads and v % 8 == address % 8
    const uint64_t amis = (uint64_t)((uintptr_t)in & 7);
    const uint8_t *base8 = in - amis;                // base8 + v is the address of v-offset v
    uint8_t *safe_byte = const_cast<uint8_t *>(base8);   // always-readable address for masked-off gather lanes
    const uint64_t limit_v = un.in_limit + amis;
    const uint64_t endbits = 8 * limit_v;
    const uint32_t *ring32 = reinterpret_cast<const uint32_t *>(L.inring);

    // ---- wave-uniform decoder state ----
    uint64_t P = 8 * (un.in_off + amis);     // bit position (v-space) of the next unread bit
    uint64_t s_hi = 0;                       // input staged: [s_hi - IN_RING, s_hi)
    Status st{SEMZ_OK, 0};
    uint32_t wsize = un.param0 ? un.param0 : 32768u;
    uint64_t prefix_amount = 0;
    const uint8_t *prefix_end = in;          // one past the last history byte
    int state = ST_HDR;
    bool final_block = false;
    bool finished = false;                   // final block fully decoded
    uint32_t stored_left = 0;
    uint32_t aux = 0;
    int dist_mode = MODE_NORMAL;
    bool lit_full = false, dist_full = false;   // ht-full-read-p, huffman.lisp:163
    int lit_max = 15, dist_max = 15;
    int nlit = 0, ndist = 0;
    const int trailing_bits = un.kind == SEMZ_UNIT_GZIP ? 64 : un.kind == SEMZ_UNIT_ZLIB ? 32 : 0;
    uint64_t pos = 0;                        // output bytes produced
    uint64_t flushed = 0;                    // output bytes already in HBM

    // ---- container header ----
    if (un.kind == SEMZ_UNIT_GZIP) {
        const uint64_t p = uni(parse_gzip_header(in, un.in_off, un.in_limit, st));
        st.code = uni(st.code); st.detail = uni(st.detail);
        aux = (uint32_t)(p - un.in_off);
        P = 8 * (p + amis);
        wsize = 32768u;
    } else if (un.kind == SEMZ_UNIT_ZLIB) {              // zlib.lisp:21-57
        uint64_t p = un.in_off;
        if (un.in_limit - p < 2) st.set(SEMZ_EOF, 0);
        else {
            const uint32_t cmf = uni((uint32_t)in[p]), flg = uni((uint32_t)in[p + 1]);
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
            P = 8 * (p + amis);
        }
    }
    if (un.prefix_len) {                                  // deflate.lisp:419-425
        prefix_amount = un.prefix_len < wsize ? un.prefix_len : wsize;
        prefix_end = in + un.prefix_off + un.prefix_len;
    }

    // ---- input staging: HBM -> LDS ring, 8 bytes per lane; what lies past the end of the input reads as zero ----
    auto stage_piece = [&]() {
        const uint64_t v = s_hi + 8u * lane;
        const uint64_t last = limit_v ? (limit_v - 1) & ~7ull : 0;
        uint2 x = *reinterpret_cast<const uint2 *>(base8 + (v < last ? v : last));
        if (v + 8 > limit_v) {
            const uint32_t nb = v >= limit_v ? 0u : (uint32_t)(limit_v - v);          // valid bytes, 0..7
            const uint64_t keep = nb ? (~0ull >> (64 - 8 * nb)) : 0ull;
            x.x &= (uint32_t)keep;
            x.y &= (uint32_t)(keep >> 32);
        }
        lds_sync();
        const uint32_t ro = (uint32_t)v & (IN_RING - 1);
        *reinterpret_cast<uint2 *>(L.inring + ro) = x;
        if (ro < 16) *reinterpret_cast<uint2 *>(L.inring + IN_RING + ro) = x;
        join_lanes();
        lds_sync();
        s_hi += IN_PIECE;
    };
    // makes [P/8, P/8 + 32) readable from the ring (a window takes at most 28 bytes from P/8 on)
    auto ensure_staged = [&]() {
        const uint64_t b = P >> 3;
        if (b >= s_hi || b + IN_RING < s_hi) s_hi = b & ~(uint64_t)(IN_PIECE - 1);      // start, or jumped out of the ring
        while (b + 32 > s_hi) stage_piece();
    };
    // 32 bits at bit position q, scalar (block headers, code lengths, the careful token step)
    auto rd32 = [&](uint64_t q) -> uint32_t {
        const uint32_t d = (uint32_t)(q >> 5);
        const uint32_t *wp = ring32 + (d & (IN_RING / 4 - 1));
        const uint32_t w0 = uni(wp[0]), w1 = uni(wp[1]);
        return (uint32_t)((((uint64_t)w1 << 32) | w0) >> (q & 31));
    };
    auto short_of = [&](uint32_t n) -> bool { return endbits - P < (uint64_t)n; };

    // A round holds up to QCAP matches.  Longer rounds mean fewer of them (each costs ~400 instructions whatever it
    // holds) and more matches whose source lies inside the round (28 instructions each, one after the other): 128
    // is worth +3.7 % on 256 KiB members, and on 4 KiB records, where every distance is short, 64 is worth +0.8 %.
    const uint32_t qcap = un.out_cap > 8192 ? (uint32_t)QCAP : (uint32_t)QCAP / 2;
    STAMP_DECL
    // (a block of assembler in the resolve step adds no base to its ring indices: the ring is this kernel's first LDS object)
    const bool ring_at_0 = (uint32_t)__builtin_amdgcn_readfirstlane(
                               (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)L.ring) == 0;
    // ---- main loop: one round per iteration ----
    for (;;) {
        // (semz_device.h: the units of a SIMD end together.  A record of a few kilobytes never asks)
        if (pos >= 8192) semz_take_turn(PRIO_TURN_SHIFT);
        STAMP(5);
        // scripts/check_uniform.sh: without this pin the optimiser's merge of P behind the lane-masked stores of the
        // round counts as divergent and the whole decoder state moves to VGPRs
        P = uni(P);
        int ev = st.code != SEMZ_OK ? EV_DONE : EV_ROUND;
        uint32_t ntok = 0, rbytes = 0;
        const uint32_t room = (uint32_t)(out_cap - pos < 0xFFFFFFFFull ? out_cap - pos : 0xFFFFFFFFull);
        const uint32_t avail = (uint32_t)(prefix_amount + pos < 0x7FFFFFFFull ? prefix_amount + pos : 0x7FFFFFFFull);
        const uint32_t rpos = (uint32_t)pos + shift;
#define FAIL(c, d) { st.set((c), (d)); ev = EV_DONE; break; }
        while (ev == EV_ROUND && ntok < qcap && rbytes < (uint32_t)ROUND_BYTES) {
            ensure_staged();
            if (state == ST_HUFF) {
                if (dist_mode != MODE_LITONLY && endbits - P >= 64 + TOKEN_BITS) {
                    bool careful = false;
                    // 32-bit stand-ins for the loop conditions: bits to the end of the input (saturated), staged bits ahead
                    const uint32_t left = (uint32_t)(endbits - P < 0x7FFFFFFFull ? endbits - P : 0x7FFFFFFFull);
                    const uint32_t sleft = (uint32_t)(8 * s_hi - P);
                    // What decides whether a window is kept whole and whether another one follows is kept as four
                    // countdowns that must all stay non-negative (one OR and one sign test), and the bit position
                    // inside the loop as its low 32 bits (all the ring index looks at):
                    //   c_tok  queue slots left after this window, minus one      c_byte  round bytes left, minus one
                    //   c_left bits to the end of the input, minus 64 + TOKEN_BITS   c_stg  staged bits ahead, minus 256
                    // `gate` has the sign bit set when windows cannot be kept whole at all in this round.
                    // (the round's byte limit is what the output still has room for when that is less: a 4 KiB record's last
                    // thousand bytes, which went through the cut path window by window, 85 windows of a record's 203)
                    const uint32_t lim = room < (uint32_t)ROUND_BYTES ? room : (uint32_t)ROUND_BYTES;
                    int32_t c_tok = (int32_t)(qcap - 1 - ntok), c_byte = (int32_t)(lim - 1 - rbytes);
                    int32_t c_left = (int32_t)(left - (64 + TOKEN_BITS)), c_stg = (int32_t)(sleft - 256);
                    const uint32_t gate = wsize >= 32768u ? 0u : 0x80000000u;
                    // (history available in front of the round; once it covers the window the test cannot fail, and it is
                    // made all the same: two vector instructions and an AND where a test of a flag costs three scalar ones)
                    const uint32_t avail_sat = avail < 32768u ? avail : 0x40000000u;
                    uint32_t Pw = (uint32_t)P;
                    const uint32_t Pw0 = Pw;
                    // what the last window of the loop leaves for the code behind it
                    uint32_t e, length, distance, ol, ooff, mrank, off;
                    uint64_t litmask, stopmask, mask, matchmask;
                    bool cut_window;                                           // the last window could not be kept whole
                    for (;;) {
                        // ---- window decode: lane l takes the token that would start at bit P + l ----
                        COUNT(0, 1);
                        const uint32_t bp = Pw + lane;                         // the ring index only needs the low bits
                        const uint32_t sh = bp & 31;
                        const uint32_t *wp = ring32 + ((bp >> 5) & (IN_RING / 4 - 1));
                        const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
                        e = L.lit_root[lo & ((1u << LIT_ROOT_BITS) - 1)];
                        const uint32_t cl = e & 15;
                        const uint32_t lx = (e >> 5) & 7;                      // 0 in literal entries
                        length = 3 + (e >> 8) + __builtin_amdgcn_ubfe(lo, cl, lx);
                        const uint32_t o2 = cl + lx;                           // <= 20
                        const uint32_t y = __builtin_amdgcn_alignbit(hi, lo, o2);
                        uint32_t dd = L.dist_root[y & ((1u << DIST_ROOT_BITS) - 1)];
                        // Lane predicates are kept as 64-bit masks in scalar registers and combined there; lane_of() turns
                        // a mask back into the predicate of a select or a branch without an instruction.
                        litmask = __builtin_amdgcn_ballot_w64((e & LIT_FLAG) != 0);
                        const uint64_t litstop = __builtin_amdgcn_ballot_w64(cl == 0);
                        const bool islit = lane_of(litmask);
                        mask = 0;
                        uint32_t offb = 0u - 64u, tlen;
                        uint32_t dl = dd & 15, dx = (dd >> 4) & 15;
                        // tokens the scalar step has to look at have a zero root entry: long codes, block end, invalid codes
                        stopmask = litstop | (__builtin_amdgcn_ballot_w64(dl == 0) & ~litmask);
                        uint32_t tbits = lane_of(stopmask) ? 64u : (islit ? cl : o2 + dl + dx);
                        // the chain of token starts inside the window
                        // The chain is four instructions per token, written out because the compiler has no way to branch
                        // on the carry of a scalar add: `off` runs biased by 2^32 - 64 (v_readlane and s_bitset1_b64 only
                        // look at its low six bits), so the add carries out exactly when the chain leaves the window.
                        // The s_nop is the wait state gfx950 wants between the VALU write of tbits and a v_readlane of it.
                        // (four steps per pass: a taken branch costs three instructions' time, DESIGN.md 3.0, and a window
                        // holds five tokens or so)
#define CHAIN_STEP "v_readlane_b32 %[t], %[tb], %[off]\n\ts_bitset1_b64 %[m], %[off]\n\ts_add_u32 %[off], %[off], %[t]\n\t"
#define CHAIN()                                                                          \
                        asm volatile("s_nop 0\n"                                         \
                                     "1:\n\t"                                            \
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"                  \
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"                  \
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"                  \
                                     CHAIN_STEP "s_cbranch_scc0 1b\n"                    \
                                     "2:"                                                \
                                     : [m] "+s"(mask), [off] "+s"(offb), [t] "=&s"(tlen) \
                                     : [tb] "v"(tbits)                                   \
                                     : "scc")
                        CHAIN();
                        // The chain has stopped at a match whose distance code is longer than the root index (one
                        // token in 180 on text; a window cut there, the scalar step and the way back into this loop
                        // cost several windows' time): the code is searched here, lane l trying length l, its entry
                        // takes the lane's place and the chain goes on from that lane.  Anything else the scalar
                        // step has to look at stays a stop.
                        uint64_t hit = mask & stopmask & ~litstop;
                        if (__builtin_expect(hit != 0, 0)) {
                            do {
                                const uint32_t c = 63u - (uint32_t)__builtin_clzll(hit);
                                const uint32_t v15 = __brev((uint32_t)__builtin_amdgcn_readlane((int)y, (int)c)) >> 17;
                                const uint32_t lim = L.dist.limit[lane & 15];
                                const uint64_t found = __builtin_amdgcn_ballot_w64(lane >= 1 && lane <= 15 && v15 < lim);
                                if (!found) break;
                                const uint32_t fl = (uint32_t)__builtin_ctzll(found);
                                const uint32_t idx = uni((uint32_t)L.dist.offs[fl]) + (v15 >> (15 - fl)) - uni((uint32_t)L.dist.first[fl]);
                                const uint32_t ddc = dist_root_entry(uni((uint32_t)L.dist_sorted[idx & 31]), fl);
                                if (ddc == 0) break;                           // (30, 31: the scalar step reports them)
                                dd = lane == c ? ddc : dd;
                                dl = dd & 15; dx = (dd >> 4) & 15;
                                stopmask = litstop | (__builtin_amdgcn_ballot_w64(dl == 0) & ~litmask);
                                tbits = lane_of(stopmask) ? 64u : (islit ? cl : o2 + dl + dx);
                                offb = c - 64u;
                                CHAIN();
                                hit = mask & stopmask & ~litstop;
                            } while (hit);
                        }
#undef CHAIN
#undef CHAIN_STEP
                        distance = 1 + (((dd >> 8) & 3) << dx) + __builtin_amdgcn_ubfe(y, dl, dx);
                        off = offb + 64u;
                        matchmask = mask & ~(stopmask | litmask);
                        // (a token of the chain that the scalar step has to take counts as more bytes than a round holds:
                        // the window is then not kept whole without a test of its own, and offsets up to that token,
                        // which is all that is used of such a window, are not affected)
                        {
                            uint32_t t0 = islit ? 1u : length;                 // (three selects, no branch)
                            t0 = lane_of(stopmask) ? 2u * ROUND_BYTES : t0;
                            ol = lane_of(mask) ? t0 : 0u;
                        }
                        const uint32_t incl = wave_scan_add(ol);
                        // output offset inside the round: bytes of the round so far (ROUND_BYTES - 1 - c_byte) + what this
                        // window puts in front of the lane's token; the uniform part is taken off on the vector side
                        ooff = ((incl - ol) - (uint32_t)c_byte) + (lim - 1);
                        mrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(matchmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)matchmask, 0));
                        const uint32_t qslot = (mrank - (uint32_t)c_tok) + (qcap - 1);   // queue slot of the lane's match
                        // the usual window: no token for the scalar step, everything fits the round with a byte and a
                        // queue slot to spare: keep it whole.  The countdowns are taken down first (the results are
                        // what the tests look at); a window that is not kept puts them back.
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        const uint32_t nm = (uint32_t)__builtin_popcountll(matchmask);
                        c_byte -= (int32_t)total;
                        c_tok -= (int32_t)nm;
                        // (all operands are far below 2^31: one sign test for all limits)
                        const int32_t fits = c_byte | c_tok | (int32_t)gate;
                        if (__builtin_expect(fits < 0, 0) ||
                            __builtin_expect((__builtin_amdgcn_ballot_w64(distance > avail_sat + ooff) & matchmask) != 0, 0)) {
                            c_byte += (int32_t)total;
                            c_tok += (int32_t)nm;
                            cut_window = true;
                            break;
                        }
                        // (every lane stores: the ones without a literal into a byte of their own of `flags`, which only
                        // the resolve step uses, and it clears them first.  A select on the vector side instead of an
                        // exec mask set up and taken down on the scalar side, which is the side this loop is short of)
                        reinterpret_cast<uint8_t *>(&L)[lane_of(mask & litmask) ? (rpos + ooff) & RING_MASK
                                                                              : (uint32_t)__builtin_offsetof(UnitLds, flags) + lane] = (uint8_t)(e >> 8);
                        // (the same for the queue: lanes without a match store into eight bytes of their own of the tables
                        // that only the construction of the next block's trees uses: clen_root, lens, clens, flags)
                        *reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(&L) +
                                                   (lane_of(matchmask) ? (uint32_t)__builtin_offsetof(UnitLds, queue) + 8u * qslot
                                                                       : (uint32_t)__builtin_offsetof(UnitLds, clen_root) + 8u * lane)) =
                            make_uint2(T_MATCH | (length << 2) | (ooff << 16), distance);
                        join_lanes();
                        COUNT(1, (uint32_t)__builtin_popcountll(mask));
                        Pw += off;
                        c_left -= (int32_t)off;
                        c_stg -= (int32_t)off;
                        // straight on to the next window while the round, the staged input and the input's end allow
                        if (__builtin_expect((fits | c_left | c_stg) < 0, 0)) { cut_window = false; break; }
                    }
                    rbytes = lim - 1 - (uint32_t)c_byte;
                    ntok = qcap - 1 - (uint32_t)c_tok;
                    P += Pw - Pw0;
                    if (cut_window) {
                        COUNT(6, 1);
                        const uint32_t have = avail + ooff;                    // deflate.lisp:329-331
                        const uint64_t hardmask = mask & (stopmask | (matchmask & __builtin_amdgcn_ballot_w64(distance > (have < wsize ? have : wsize))) |
                                                          __builtin_amdgcn_ballot_w64((uint64_t)ooff + ol > (uint64_t)room));
                        const uint64_t softmask = mask & (__builtin_amdgcn_ballot_w64(ooff >= (uint32_t)ROUND_BYTES) |
                                                          (matchmask & __builtin_amdgcn_ballot_w64(ntok + mrank >= qcap)));
                        const uint64_t cutmask = hardmask | softmask;
                        uint64_t keep = mask;
                        uint32_t adv = off;
                        if (cutmask) {
                            const uint32_t c = (uint32_t)__builtin_ctzll(cutmask);
                            keep = mask & ((1ull << c) - 1);
                            adv = c;
                            careful = (hardmask >> c) & 1;
                            if (careful) COUNT(7, 1);
                            rbytes = (uint32_t)__builtin_amdgcn_readlane((int)ooff, (int)c);
                        } else {
                            rbytes = (uint32_t)__builtin_amdgcn_readlane((int)(ooff + ol), 63);
                        }
                        if (lane_of(keep & litmask)) L.ring[(rpos + ooff) & RING_MASK] = (uint8_t)(e >> 8);
                        join_lanes();
                        if (lane_of(keep & ~litmask)) L.queue[ntok + mrank] = make_uint2(T_MATCH | (length << 2) | (ooff << 16), distance);
                        join_lanes();
                        ntok += (uint32_t)__builtin_popcountll(matchmask & keep);
                        COUNT(1, (uint32_t)__builtin_popcountll(keep));
                        P += adv;
                    }
                    // the scalar step may queue a match and write a literal: only inside the round's limits
                    if (!careful || ntok >= qcap || rbytes >= (uint32_t)ROUND_BYTES) continue;
                    ensure_staged();
                }
                // ---- one token by the scalar step with every check (decode-huffman-data, deflate.lisp:313-338) ----
                {
                    COUNT(2, 1);
                    uint32_t x = rd32(P);
                    uint32_t e = uni((uint32_t)L.lit_root[x & ((1u << LIT_ROOT_BITS) - 1)]);
                    uint32_t cl = e & 15;
                    if (cl == 0) {
                        uint32_t v15 = __brev(x) >> 17, idx = 0;
                        cl = canon_find(L.lit, v15, 1, 15, idx);
                        e = lit_entry(uni((uint32_t)L.lit_sorted[idx]), cl);
                    }
                    if (short_of(lit_full ? (uint32_t)lit_max : cl)) FAIL(SEMZ_EOF, 0)
                    P += cl;
                    if (e & LIT_FLAG) {
                        if (rbytes >= room) FAIL(SEMZ_E_OUT_CAP, 0)
                        if (lane == 0) L.ring[(rpos + rbytes) & RING_MASK] = (uint8_t)(e >> 8);
                        join_lanes();
                        rbytes++;
                        continue;
                    }
                    const uint32_t lx = (e >> 5) & 7;
                    if (lx == 7) {
                        if ((e >> 8) == 0) { state = ST_HDR; continue; }
                        FAIL(SEMZ_E_DEFLATE_LITLEN_CODE, 257 + (e >> 8))
                    }
                    if (dist_mode == MODE_LITONLY) {
                        uint32_t sym = 257;
                        for (uint32_t i = 0; i < 29; i++) { uint32_t b, xx; length_code(i, b, xx); if (b == 3 + (e >> 8)) sym = 257 + i; }
                        FAIL(SEMZ_E_DEFLATE_LEN_IN_LITONLY, sym)
                    }
                    if (short_of(lx)) FAIL(SEMZ_EOF, 0)
                    x = rd32(P);
                    const uint32_t length = 3 + (e >> 8) + (x & ((1u << lx) - 1));
                    P += lx;
                    x >>= lx;                                                  // lx <= 5: 27 bits left, the code needs 15
                    uint32_t dd = uni((uint32_t)L.dist_root[x & ((1u << DIST_ROOT_BITS) - 1)]);
                    uint32_t dl = dd & 15;
                    if (dl == 0) {
                        uint32_t v15 = __brev(x) >> 17, idx = 0;
                        dl = canon_find(L.dist, v15, 1, 15, idx);
                        dd = dist_entry(uni((uint32_t)L.dist_sorted[idx]), dl);
                    }
                    if (short_of(dist_full ? (uint32_t)dist_max : dl)) FAIL(SEMZ_EOF, 0)
                    P += dl;
                    if (dd & DIST_BAD) FAIL(dist_mode == MODE_SINGLE ? SEMZ_E_DEFLATE_SINGLE_DIST : SEMZ_E_DEFLATE_DIST_CODE, dd >> 11)
                    const uint32_t dx = (dd >> 4) & 15;
                    if (short_of(dx)) FAIL(SEMZ_EOF, 0)
                    x = rd32(P);
                    const uint32_t distance = 1 + (((dd >> 8) & 3) << dx) + (x & ((1u << dx) - 1));
                    P += dx;
                    const uint32_t have = avail + rbytes;                      // deflate.lisp:329-331
                    if (distance > (have < wsize ? have : wsize)) FAIL(SEMZ_E_DEFLATE_DISTANCE, distance)
                    if (rbytes + length > room) FAIL(SEMZ_E_OUT_CAP, 0)
                    if (lane == 0) L.queue[ntok] = make_uint2(T_MATCH | (length << 2) | (rbytes << 16), distance);
                    join_lanes();
                    ntok++;
                    rbytes += length;
                    continue;
                }
            }
            if (state == ST_HDR) {
                if (final_block) { finished = true; ev = EV_DONE; break; }
                // BFINAL, BTYPE: deflate.lisp:374-397
                uint32_t x = rd32(P);
                if (short_of(1)) FAIL(SEMZ_EOF, 0)
                final_block = x & 1; P += 1;
                if (short_of(2)) FAIL(SEMZ_EOF, 0)
                const uint32_t type = (x >> 1) & 3; P += 2;
                if (type == 0) {
                    P = (P + 7) & ~7ull;
                    x = rd32(P);
                    if (short_of(16)) FAIL(SEMZ_EOF, 0)
                    const uint32_t len = x & 0xFFFF; P += 16;
                    if (short_of(16)) FAIL(SEMZ_EOF, 0)
                    const uint32_t chk = x >> 16; P += 16;
                    if (chk != (len ^ 0xFFFFu)) FAIL(SEMZ_E_DEFLATE_STORED_LEN, chk)
                    stored_left = len;
                    if (len) {
                        state = ST_STORED;
                        // the whole stored block must exist (decode-uncompressed-data, deflate.lisp:350)
                        if ((uint64_t)rbytes + len > room) FAIL(SEMZ_E_OUT_CAP, 0)
                        if (endbits - P < (uint64_t)8 * len) FAIL(SEMZ_EOF, 0)
                    }
                } else if (type == 1) { ev = EV_FIXED; break; }
                else if (type == 2) {
                    // read-dynamic-dhi, deflate.lisp:160-175: counts and code-length code lengths
                    x = rd32(P);
                    if (short_of(5)) FAIL(SEMZ_EOF, 0)
                    nlit = 257 + (int)(x & 31); P += 5;
                    if (short_of(5)) FAIL(SEMZ_EOF, 0)
                    ndist = 1 + (int)((x >> 5) & 31); P += 5;
                    if (short_of(4)) FAIL(SEMZ_EOF, 0)
                    const int nclen = 4 + (int)((x >> 10) & 15); P += 4;
                    if (lane < 19) L.clens[lane] = 0;
                    join_lanes();
                    lds_sync();
                    bool eof = false;
                    for (int i = 0; i < nclen; i++) {
                        // code-length order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (deflate.lisp:174)
                        const uint64_t ORDER_MID = 0x0B050A0609070800ull;   // entries 3..10
                        uint32_t sym;
                        if (i < 3) sym = 16 + i;
                        else if (i < 11) sym = (uint32_t)(ORDER_MID >> (8 * (i - 3))) & 0xFF;
                        else sym = (i & 1) ? (uint32_t)(19 - i) / 2u : (uint32_t)(i + 12) / 2u;
                        if ((i & 7) == 0) x = rd32(P);                       // 8 x 3 bits per read
                        if (short_of(3)) { eof = true; break; }
                        if (lane == 0) L.clens[sym] = (uint8_t)(x & 7);
                        join_lanes();
                        x >>= 3;
                        P += 3;
                    }
                    if (eof) FAIL(SEMZ_EOF, 0)
                    ev = EV_DYN;
                    break;
                } else FAIL(SEMZ_E_DEFLATE_BTYPE, 0)
                continue;
            }
            // ---- ST_STORED: raw bytes; P is byte aligned, the span is copied when the round is resolved ----
            {
                const uint32_t room_r = ROUND_BYTES - rbytes;
                const uint32_t n = stored_left < room_r ? stored_left : room_r;
                const uint64_t src = P >> 3;                                  // v-offset of the span
                if (lane == 0) L.queue[ntok] = make_uint2(T_RAW | (n << 2) | (rbytes << 16), (uint32_t)(src - amis - un.in_off));
                join_lanes();
                ntok++;
                rbytes += n;
                stored_left -= n;
                P += 8ull * n;
                if (stored_left == 0) state = ST_HDR;
            }
        }
#undef FAIL
        lds_sync();
        STAMP(0);
        COUNT(3, 1);

        // ---- resolve the round's tokens ----
        const uint64_t pos0 = pos, pos1 = pos + rbytes;
        // positions >= near_lo live in the ring; older ones are in HBM (flushed), negative ones in the preset dictionary
        const int64_t near_lo = (int64_t)pos1 > RING ? (int64_t)pos1 - RING : 0;
        if (ntok) {
            // (lane l holds tokens l and 64 + l; the second half only when the round has that many)
            uint2 tk[2];
            bool indep[2];
            uint32_t tdoff[2], poff[2];
            int32_t tsrel[2];
            uint64_t nearm[2] = {0, 0}, rawm[2] = {0, 0}, indepm[2] = {0, 0};
            uint32_t PB = 0;
            // A match whose whole source lies before the round's first byte depends on nothing produced in this
            // round (the round's literals are already in the ring).  All such matches are resolved together with
            // lanes = bytes: an exclusive prefix sum of their lengths lays their bytes out back to back, and for
            // every 64 of those bytes a lane finds its match (below), takes that match's source and destination
            // from LDS and loads its byte -- from the ring, from HBM (older than the ring: written by this wave,
            // read back through L2) or from the preset dictionary.  Four such 64-byte chunks are in flight per
            // pass.  The other matches read bytes of this round and follow, in token order.
            auto take_half = [&](const int h) {
                tk[h] = lane + 64u * h < ntok ? L.queue[lane + 64u * h] : make_uint2(0, 0);
                const bool is_match = (tk[h].x & 3) == T_MATCH;
                const uint32_t tlen = (tk[h].x >> 2) & 0x3FFF;
                tdoff[h] = tk[h].x >> 16;
                tsrel[h] = (int32_t)tdoff[h] - (int32_t)tk[h].y;        // source position relative to pos0
                indep[h] = is_match && tsrel[h] + (int32_t)tlen <= 0;
                const uint32_t plen = indep[h] ? tlen : 0u;
                const uint32_t pincl = wave_scan_add(plen);
                poff[h] = PB + pincl - plen;
                PB += (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);
                nearm[h] = __builtin_amdgcn_ballot_w64(is_match && !indep[h]);
                rawm[h] = __builtin_amdgcn_ballot_w64((tk[h].x & 3) == T_RAW);
                indepm[h] = __builtin_amdgcn_ballot_w64(indep[h]);
            };
            take_half(0);
            tk[1] = make_uint2(0, 0); indep[1] = false; tdoff[1] = 0; poff[1] = 0; tsrel[1] = 0;
            if (ntok > 64) take_half(1);
            COUNT(4, (uint32_t)__builtin_popcountll(indepm[0]) + (uint32_t)__builtin_popcountll(indepm[1]));
            COUNT(5, (uint32_t)__builtin_popcountll(nearm[0]) + (uint32_t)__builtin_popcountll(nearm[1]));
            if (PB) {
                // Which match a byte belongs to: the starts of the matches' byte ranges are bits of a bitmap over the
                // PB bytes (one LDS atomic per match and round), and a byte's match is the r-th of them, r = the
                // number of starts at or below the byte -- the starts below the chunk, counted as the chunks go
                // by, plus a popcount of the chunk's 64 bits under the lane's mask.  The matches' fields wait in
                // LDS in that order, packed into a word: A = source - byte index and B = destination - byte index,
                // both relative to the round's first byte (A in [-35 k, 0), B within +-2.3 k).  (The first version
                // scattered lane numbers into 64 flag bytes per chunk, max-scanned them and fetched three fields
                // by bpermute: 60 instructions and three LDS round trips per chunk where this takes half.)
                uint32_t *const bitmap = reinterpret_cast<uint32_t *>(L.lens);          // 320 bytes: 2,305 bits at most
                uint32_t *const packed = reinterpret_cast<uint32_t *>(L.queue);         // (the queue is in registers by now)
                lds_sync();
                if (lane < 40) reinterpret_cast<uint2 *>(bitmap)[lane] = make_uint2(0, 0);
                join_lanes();
                lds_sync();
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t irank = (h ? (uint32_t)__builtin_popcountll(indepm[0]) : 0u) +
                                           __builtin_amdgcn_mbcnt_hi((uint32_t)(indepm[h] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)indepm[h], 0));
                    if (indep[h]) {
                        __hip_atomic_fetch_or(&bitmap[poff[h] >> 5], 1u << (poff[h] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        packed[irank] = ((uint32_t)(tsrel[h] - (int32_t)poff[h]) & 0x3FFFFu) | ((uint32_t)((int32_t)tdoff[h] - (int32_t)poff[h]) << 18);
                    }
                    join_lanes();
                }
                join_lanes();
                lds_sync();
                // the lane's mask of the positions at or below it, as two words
                const uint32_t le_lo = lane < 32 ? (0xFFFFFFFFu >> (31 - lane)) : 0xFFFFFFFFu;
                const uint32_t le_hi = lane < 32 ? 0u : (0xFFFFFFFFu >> (63 - lane));
                // positions relative to the round's first byte, in 32 bits: what lies before the output (history) and
                // what is still in the ring
                const int32_t npos = pos0 < 0x7FFFFFFFull ? -(int32_t)pos0 : INT32_MIN;
                const int32_t nl_rel = (int32_t)(near_lo - (int64_t)pos0);
                const uint32_t rp0 = (uint32_t)pos0 + shift;
                // (loads from HBM go through one uniform base and a 32-bit offset; lanes with nothing to fetch
                // there read a byte that exists: the last flushed one, or the unit's first)
                constexpr int32_t GBIAS = 1 << 16;
                const uint8_t *const gbase = reinterpret_cast<const uint8_t *>((uintptr_t)(ob + pos0) - (uintptr_t)GBIAS);
                const int32_t safe_rel = near_lo > 0 ? nl_rel - 1 : npos;
                const bool far_possible = near_lo > 0;
                const bool may_neg = prefix_amount != 0 && pos0 < 40000;        // sources in the preset history at all?
                uint32_t before = 0;                // starts in front of the chunk
                // (a pass of NC chunks, NC = 4, 2 or 1: the last pass of a round is as wide as what is left)
                auto pass = [&](const uint32_t cb, auto nc_tag) {
                    constexpr int NC = decltype(nc_tag)::value;
                    // A pass of four chunks without a branch (bits beyond PB are zero, lanes beyond PB are masked at
                    // the end), in three steps so that the LDS round trips of the four chunks overlap: the counts,
                    // the matches' words, the bytes.
                    uint32_t r[NC], w[NC], gv[NC], ov[NC], di[NC];
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const uint2 sb = *reinterpret_cast<const uint2 *>(bitmap + ((cb + 64u * c) >> 5));
                        r[c] = (uint32_t)__builtin_popcount(sb.x & le_lo) + (uint32_t)__builtin_popcount(sb.y & le_hi);
                    }
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        w[c] = packed[(before + r[c] - 1) & (QCAP - 1)];
                        before += (uint32_t)__builtin_amdgcn_readlane((int)r[c], 63);
                    }
                    uint64_t negm = 0;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const uint32_t bidx = cb + 64u * c + lane;                   // the byte's index among the PB bytes
                        const int32_t srel = ((int32_t)(w[c] << 14) >> 14) + (int32_t)bidx;    // source, relative to pos0: negative
                        const uint32_t drel = (uint32_t)(((int32_t)w[c] >> 18) + (int32_t)bidx);
                        const bool act = bidx < PB;
                        const bool useg = act && srel >= npos && srel < nl_rel;
                        // (while the unit's output still fits the ring nothing is older than the ring: no loads at
                        // all -- a 4 KiB record never makes one, and a load that nobody needs is still waited for)
                        gv[c] = far_possible ? __hip_atomic_load(gbase + (uint32_t)((useg ? srel : safe_rel) + GBIAS), __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT)
                                             : 0u;
                        ov[c] = L.ring[(rp0 + (uint32_t)srel) & RING_MASK];
                        di[c] = act ? ((rp0 + drel) & RING_MASK) | (useg ? 0x80000000u : 0u) : 0xFFFFFFFFu;
                        if (may_neg) negm |= __builtin_amdgcn_ballot_w64(act && srel < npos);
                    }
                    if (may_neg && negm) {                  // preset dictionary bytes (zlib FDICT, deflate prefix)
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            const uint32_t bidx = cb + 64u * c + lane;
                            const int32_t srel = ((int32_t)(w[c] << 14) >> 14) + (int32_t)bidx;
                            if (bidx < PB && srel < npos) ov[c] = prefix_end[(int64_t)pos0 + srel];
                            join_lanes();
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (di[c] != 0xFFFFFFFFu) L.ring[di[c] & RING_MASK] = (uint8_t)((di[c] >> 31) ? gv[c] : ov[c]);
                        join_lanes();
                    }
                };
                for (uint32_t cb = 0; cb < PB;) {
                    const uint32_t left = PB - cb;
                    if (left > 128) { pass(cb, std::integral_constant<int, 4>{}); cb = uni(cb + 256); }
                    else if (left > 64) { pass(cb, std::integral_constant<int, 2>{}); cb = uni(cb + 128); }
                    else { pass(cb, std::integral_constant<int, 1>{}); cb = uni(cb + 64); }
                }
            }
            // stored spans straight from the input
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
              uint64_t rm = h ? rawm[1] : rawm[0];
              const uint32_t vx = h ? tk[1].x : tk[0].x, vy = h ? tk[1].y : tk[0].y;
              while (rm) {
                const uint32_t j = (uint32_t)__builtin_ctzll(rm);
                rm &= rm - 1;
                const uint32_t tx = (uint32_t)__builtin_amdgcn_readlane((int)vx, (int)j);
                const uint32_t ty = (uint32_t)__builtin_amdgcn_readlane((int)vy, (int)j);
                const uint32_t len = (tx >> 2) & 0x3FFF;
                const uint32_t p = (uint32_t)(pos0 + (tx >> 16)) + shift;
                const uint8_t *s = in + un.in_off + ty;
                for (uint32_t k0 = 0; k0 < len; k0 = uni(k0 + 64))
                    if (k0 + lane < len) L.ring[(p + k0 + lane) & RING_MASK] = s[k0 + lane];
              }
            }
            // near matches in token order (the lanes' first tokens, then their second ones); LDS operations of one
            // wave execute in order
            // (what the round has in front of it, saturated: a source can only begin in the preset history while this is small)
            const uint32_t hist32 = pos0 < 0x7FFFFFFFull ? (uint32_t)pos0 : 0x7FFFFFFFu;
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
              uint64_t nm = h ? nearm[1] : nearm[0];
              const uint32_t vx = h ? tk[1].x : tk[0].x, vy = h ? tk[1].y : tk[0].y;
              while (nm) {
                if (ring_at_0) {
                    // The usual ones (at most 64 bytes, at least as far back as long, the source in the output) one
                    // after the other without leaving the block: 26 instructions and one taken branch a match where
                    // the compiler's version of the same path has 45 and four, each condition kept as a flag
                    // register first.  Leaves in front of the first match that is not of this kind.
                    static_assert(RING_MASK == 0xfff, "the ring mask is written out in the block below");
                    uint64_t ex;
                    uint32_t sj, stx, sd, slen, sdoff, st0, sp, va, vb, vv;
                    asm volatile("s_mov_b64 %[ex], exec\n"
                                 "1:\n\t"
                                 "s_ff1_i32_b64 %[j], %[nm]\n\t"
                                 "v_readlane_b32 %[tx], %[vx], %[j]\n\t"
                                 "v_readlane_b32 %[d], %[vy], %[j]\n\t"
                                 "s_bfe_u32 %[len], %[tx], 0xe0002\n\t"
                                 "s_lshr_b32 %[doff], %[tx], 16\n\t"
                                 "s_cmp_gt_u32 %[len], 64\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_cmp_lt_u32 %[d], %[len]\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_add_u32 %[t], %[hist], %[doff]\n\t"
                                 "s_cmp_gt_u32 %[d], %[t]\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_add_u32 %[p], %[base], %[doff]\n\t"
                                 "s_sub_u32 %[t], %[p], %[d]\n\t"
                                 "v_add_u32_e32 %[a], %[t], %[lane]\n\t"
                                 "v_and_b32_e32 %[a], 0xfff, %[a]\n\t"
                                 "ds_read_u8 %[v], %[a]\n\t"
                                 "v_add_u32_e32 %[b], %[p], %[lane]\n\t"
                                 "v_and_b32_e32 %[b], 0xfff, %[b]\n\t"
                                 "s_bitset0_b64 %[nm], %[j]\n\t"
                                 "v_cmpx_gt_u32_e32 vcc, %[len], %[lane]\n\t"
                                 "s_waitcnt lgkmcnt(0)\n\t"
                                 "ds_write_b8 %[b], %[v]\n\t"
                                 "s_mov_b64 exec, %[ex]\n\t"
                                 "s_cmp_lg_u64 %[nm], 0\n\t"
                                 "s_cbranch_scc1 1b\n"
                                 "2:"
                                 : [nm] "+s"(nm), [ex] "=&s"(ex), [j] "=&s"(sj), [tx] "=&s"(stx), [d] "=&s"(sd), [len] "=&s"(slen),
                                   [doff] "=&s"(sdoff), [t] "=&s"(st0), [p] "=&s"(sp), [a] "=&v"(va), [b] "=&v"(vb), [v] "=&v"(vv)
                                 : [vx] "v"(vx), [vy] "v"(vy), [lane] "v"(lane), [hist] "s"(hist32), [base] "s"((uint32_t)pos0 + shift)
                                 : "scc", "vcc", "memory");
                    if (!nm) break;
                }
                const uint32_t j = (uint32_t)__builtin_ctzll(nm);
                nm &= nm - 1;
                const uint32_t tx = (uint32_t)__builtin_amdgcn_readlane((int)vx, (int)j);
                const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)vy, (int)j);
                const uint32_t len = (tx >> 2) & 0x3FFF;
                const uint32_t p = (uint32_t)(pos0 + (tx >> 16)) + shift;     // ring index of the destination
                const uint32_t src = p - dist;
                // the usual one first: at most 64 bytes, at least as far back as it is long, the source in the output
                if (__builtin_expect(len <= 64 && dist >= len && dist <= hist32 + (tx >> 16), 1)) {
                    const uint32_t v = L.ring[(src + lane) & RING_MASK];
                    if (lane < len) L.ring[(p + lane) & RING_MASK] = (uint8_t)v;
                    join_lanes();
                    continue;
                }
                const int64_t sabs0 = (int64_t)(pos0 + (tx >> 16)) - (int64_t)dist;
                if (sabs0 < 0) {
                    // the source starts in the preset history and runs on into this round's output: byte k repeats
                    // byte k mod dist of the window [sabs0, sabs0 + dist), which is final (history, or the ring)
                    for (uint32_t k0 = 0; k0 < len; k0 = uni(k0 + 64)) {
                        const uint32_t k = k0 + lane;
                        const int64_t s = sabs0 + (int64_t)(k % dist);
                        const uint32_t v = s < 0 ? (uint32_t)prefix_end[k < len ? s : -1]
                                                 : (uint32_t)L.ring[(uint32_t)((uint64_t)s + shift) & RING_MASK];
                        if (k < len) L.ring[(p + k) & RING_MASK] = (uint8_t)v;
                    }
                    join_lanes();
                } else if (dist >= len) {
                    for (uint32_t k0 = 0; k0 < len; k0 = uni(k0 + 64)) {
                        const uint32_t k = k0 + lane;
                        const uint32_t v = L.ring[(src + k) & RING_MASK];
                        if (k < len) L.ring[(p + k) & RING_MASK] = (uint8_t)v;
                    }
                } else if (dist == 1) {
                    const uint32_t v = L.ring[src & RING_MASK];
                    for (uint32_t k0 = 0; k0 < len; k0 = uni(k0 + 64))
                        if (k0 + lane < len) L.ring[(p + k0 + lane) & RING_MASK] = (uint8_t)v;
                } else {
                    // periodic source: the first `dist` bytes are final, every later byte repeats them
                    for (uint32_t k0 = 0; k0 < len; k0 = uni(k0 + 64)) {
                        const uint32_t k = k0 + lane;
                        const uint32_t v = L.ring[(src + k % dist) & RING_MASK];
                        if (k < len) L.ring[(p + k) & RING_MASK] = (uint8_t)v;
                    }
                }
              }
            }
        }
        pos = pos1;
        lds_sync();
        STAMP(2);

        // ---- flush the ring ----
        {
            const bool last = ev == EV_DONE;
            const uint32_t head = (16 - shift) & 15;
            if (flushed == 0 && head && (pos >= head || last)) {
                const uint32_t n = pos < head ? (uint32_t)pos : head;
                if (lane < n) ob[lane] = L.ring[(lane + shift) & RING_MASK];
                join_lanes();
                flushed = n;
            }
            flushed = uni(flushed);      // the optimiser threads the lane-masked store into this update: pin it as uniform
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
                    join_lanes();
                    flushed = pos;
                }
            }
            flushed = uni(flushed);
        }
        lds_sync();
        STAMP(3);

        // ---- table construction for the next block ----
        if (ev == EV_DYN || ev == EV_FIXED) {
            long overread = 0;
            if (ev == EV_DYN) {
                // code-length tree, then the code lengths themselves (deflate.lisp:176-197)
                TreeMeta &clen_meta = *reinterpret_cast<TreeMeta *>(L.lens + CLEN_META_AT);
                build_tree<CLEN_ROOT_BITS>(L.clens, 19, clen_meta, L.lens + CLEN_SORTED_AT, L.clen_root, clen_entry);
                const int tstatus = (int)uni(clen_meta.status);
                overread = (final_block ? 0 : MIN_BLOCK_BITS) + trailing_bits;    // deflate.lisp:392-396
                if (tstatus == SEMZ_OK) {
                    const uint32_t cminl = uni(clen_meta.minlen), cmaxl = uni(clen_meta.maxlen);
                    const bool cfull = overread >= (long)(cmaxl - cminl);
                    const int total = nlit + ndist;
                    int i = 0;
                    uint32_t prev = 0;
                    uint32_t x = 0, have = 0;                // the next `have` bits of the input from P on
                    // Windows first, as for the tokens: lane l decodes the code that would start at bit P + l, the
                    // chain of real starts is followed with readlane, a prefix sum over the run lengths places every
                    // code, and a repeat (16) takes the value of the nearest code below it that is not one.  Runs of
                    // zeros (17, 18) store nothing: the array starts out zeroed.  A window that holds any of the
                    // reference's failures, or comes near the end of the input, is left to the loop below.
                    for (uint32_t k = lane; k < sizeof L.lens; k += 64) L.lens[k] = 0;
                    join_lanes();
                    lds_sync();
                    while (i < total && endbits - P >= 64 + 14 + 64) {
                        ensure_staged();
                        const uint32_t bp = (uint32_t)P + lane;
                        const uint32_t *wp = ring32 + ((bp >> 5) & (IN_RING / 4 - 1));
                        const uint32_t wx = __builtin_amdgcn_alignbit(wp[1], wp[0], bp & 31);
                        const uint32_t ce = L.clen_root[wx & ((1u << CLEN_ROOT_BITS) - 1)];
                        const uint32_t cl = ce & 7, sym = ce >> 3;
                        const uint32_t xb = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
                        const uint32_t n = sym < 16 ? 1u : (sym == 18 ? 11u : 3u) + __builtin_amdgcn_ubfe(wx, cl, xb);
                        const uint32_t tb = (cl ? cl : 1u) + xb;          // (a complete tree has no empty entry; the chain must end all the same)
                        // (the chain as in the token windows: `offb` biased by 2^32 - 64, the add carries out at the end)
                        uint64_t starts = 0;
                        uint32_t offb = 0u - 64u, tlen;
#define CHAIN_STEP "v_readlane_b32 %[t], %[tb], %[off]\n\ts_bitset1_b64 %[m], %[off]\n\ts_add_u32 %[off], %[off], %[t]\n\t"
                        asm volatile("s_nop 0\n"
                                     "1:\n\t"
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"
                                     CHAIN_STEP "s_cbranch_scc1 2f\n\t"
                                     CHAIN_STEP "s_cbranch_scc0 1b\n"
                                     "2:"
                                     : [m] "+s"(starts), [off] "+s"(offb), [t] "=&s"(tlen)
                                     : [tb] "v"(tb)
                                     : "scc");
#undef CHAIN_STEP
                        const bool isstart = lane_of(starts);
                        const uint32_t nm = isstart ? n : 0u;
                        const uint32_t posl = (uint32_t)i + wave_scan_add(nm) - nm;
                        const uint64_t um = __builtin_amdgcn_ballot_w64(isstart && posl < (uint32_t)total);
                        const bool usable = lane_of(um);
                        if (__builtin_amdgcn_ballot_w64(usable && (cl == 0 || posl + n > (uint32_t)total || (sym == 16 && posl == 0)))) break;
                        const uint64_t m16 = __builtin_amdgcn_ballot_w64(usable && sym == 16);
                        const uint64_t lower = um & ~m16 & ((1ull << lane) - 1);
                        const uint32_t own = sym < 16 ? sym : 0u;
                        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (63 - (int)__builtin_clzll(lower | 1)), (int)own);
                        const uint32_t v = sym == 16 ? (lower ? got : prev) : own;
                        if (usable && sym < 16) L.lens[posl] = (uint8_t)sym;
                        join_lanes();
                        if (m16) {
                            for (uint32_t k = 0; k < 6; k++)
                                if (lane_of(m16) && k < n) L.lens[posl + k] = (uint8_t)v;
                            join_lanes();
                        }
                        const int last = 63 - (int)__builtin_clzll(um);
                        prev = (uint32_t)__builtin_amdgcn_readlane((int)v, last);
                        i = __builtin_amdgcn_readlane((int)(posl + n), last);
                        P += (uint32_t)last + (uint32_t)__builtin_amdgcn_readlane((int)tb, last);
                    }
                    while (i < total) {
                        if (have < 14) {                     // a code takes at most 7 bits and 7 extra bits
                            ensure_staged();
                            x = rd32(P);
                            have = 32;
                        }
                        const uint32_t ce = uni((uint32_t)L.clen_root[x & ((1u << CLEN_ROOT_BITS) - 1)]);
                        const uint32_t cl = ce & 7;
                        if (short_of(cfull ? cmaxl : cl)) { st.set(SEMZ_EOF, 0); break; }
                        P += cl;
                        x >>= cl;
                        have -= cl;
                        const uint32_t sym = ce >> 3;
                        if (sym < 16) {
                            L.lens[i] = (uint8_t)sym;        // every lane, same address, same value
                            prev = sym;
                            i++;
                            continue;
                        }
                        const uint32_t xb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                        if (short_of(xb)) { st.set(SEMZ_EOF, 0); break; }
                        const int n = (sym == 18 ? 11 : 3) + (int)(x & ((1u << xb) - 1));
                        P += xb;
                        x >>= xb;
                        have -= xb;
                        if (i + n > total) { st.set(SEMZ_E_DEFLATE_CODELEN_BOUNDS, 0); break; }
                        uint32_t v = 0;
                        if (sym == 16) {
                            if (i == 0) { st.set(SEMZ_E_DEFLATE_REPEAT_FIRST, 0); break; }
                            v = prev;
                        }
                        for (int k0 = 0; k0 < n; k0 = uni(k0 + 64))
                            if (k0 + (int)lane < n) L.lens[i + k0 + lane] = (uint8_t)v;
                        join_lanes();
                        prev = v;
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
                build_tree<LIT_ROOT_BITS>(L.lens, nlit, L.lit, L.lit_sorted, L.lit_root, lit_root_entry);
                const int lstatus = (int)uni(L.lit.status);
                if (lstatus != SEMZ_OK) st.set(lstatus, 0);
                else {
                    // lengths->dist-ht special cases, deflate.lisp:91-122
                    int mode = MODE_NORMAL;
                    const uint8_t *dlens = L.lens + nlit;
                    if (ndist == 1 && uni((uint32_t)dlens[0]) == 0) mode = MODE_LITONLY;
                    else {
                        // one code of length one and nothing else?
                        const uint32_t v = (int)lane < ndist ? dlens[lane] : 0u;
                        const uint64_t nz = __builtin_amdgcn_ballot_w64(v != 0), ones = __builtin_amdgcn_ballot_w64(v == 1);
                        if (__builtin_popcountll(nz) == 1 && nz == ones) mode = MODE_SINGLE | ((int)__builtin_ctzll(nz) << 8);
                    }
                    if ((mode & 255) != MODE_NORMAL) {
                        // synthetic 32-entry length vector, reusing the lit/len part of lens[] (already consumed)
                        lds_sync();
                        if (lane < 32) {
                            uint8_t v = 0;
                            if ((mode & 255) == MODE_LITONLY) v = lane >= 30;
                            else { const int one = mode >> 8; v = (int)lane == one || (int)lane == (one == 31 ? 30 : 31); }
                            L.lens[lane] = v;
                        }
                        lds_sync();
                        build_tree<DIST_ROOT_BITS>(L.lens, 32, L.dist, L.dist_sorted, L.dist_root, dist_root_entry);
                    } else {
                        build_tree<DIST_ROOT_BITS>(dlens, ndist, L.dist, L.dist_sorted, L.dist_root, dist_root_entry);
                    }
                    const int dstatus = (int)uni(L.dist.status);
                    if (dstatus != SEMZ_OK) st.set(dstatus, 0);
                    dist_mode = mode & 255;
                    const uint32_t lmin = uni(L.lit.minlen), lmax = uni(L.lit.maxlen), dmin = uni(L.dist.minlen), dmax = uni(L.dist.maxlen);
                    lit_max = (int)lmax; dist_max = (int)dmax;
                    lit_full = overread >= (long)(lmax - lmin);
                    dist_full = overread >= (long)(dmax - dmin);
                    state = ST_HUFF;
                }
            }
            lds_sync();
        }

        STAMP(4);

        // ---- unit end: trailer and results ----
        if (ev == EV_DONE) {
            semz_tail t{};
            semz_result r{};
            if (st.code == SEMZ_OK && finished) {
                // the trailer is byte aligned: read it straight from memory
                const uint64_t bp = ((P + 7) >> 3) - amis;            // offset inside `in`
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
                if (un.kind != SEMZ_UNIT_DEFLATE) P = 8 * (endb + amis);
            }
            t.trailer_end_bit = P - 8 * amis;
            r.status = st.code;
            r.detail = st.detail;
            r.in_end_bit = P - 8 * amis;
            r.out_len = pos;
            r.out_off = un.out_off;
            r.aux = aux;
            if (lane == 0) {
                results[u] = r;
                tails[u] = t;
#ifdef SEMZ_STAMPS
                for (int i = 0; i < 6; i++) atomicAdd(&g_stamps[i], (unsigned long long)t_acc[i]);
                for (int i = 0; i < 8; i++) atomicAdd(&g_stamps[6 + i], (unsigned long long)n_acc[i]);
                atomicAdd(&g_stamps[14], 1ull);
                if (u < 16384) {
                    g_unit_times[u][0] = t_unit0;
                    g_unit_times[u][1] = __builtin_amdgcn_s_memrealtime();
                    g_unit_times[u][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 20);
                }
#endif
            }
            break;
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

#ifdef SEMZ_STAMPS
extern "C" int semz_debug_unit_times(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_unit_times), (size_t)n * 24) == hipSuccess ? 0 : -1;
}
#endif

extern "C" void semz_launch_inflate(const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                                    uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails,
                                    hipStream_t stream)
{
    if (nunits == 0) return;
    inflate_kernel<<<nunits, 64, 0, stream>>>(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails);
}
