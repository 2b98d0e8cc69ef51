// lzma2.hip -- gfx950 kernels for xz blocks and raw LZMA2 streams.
//
// Replaces, per unit, the reference loops
//   lzma-range-decoder, lrd-*            lzma.lisp:49-198
//   decode-lzma-length / -distance       lzma.lisp:247-341
//   decode-lzma (all eof modes)          lzma.lisp:566-775
//   :raw-lzma chunk method               lzma.lisp:867-895
//   LZMA2 control-byte state machine     lzma2.lisp:77-204
//   xz block end (sizes, padding, check) xz.lisp:426-457
//   x86 BCJ filter                       xz-bcj-filters.lisp:169-219
//   ARM/ARM64/Thumb/IA-64/PowerPC/SPARC  xz-bcj-filters.lisp:45-167
//   delta filter                         xz.lisp:238-258
//
// lzma2_kernel: one unit per 64-lane wavefront (one wavefront per workgroup).
// LZMA is bit-serial with adaptive probabilities, so the symbol loop is
// wave-uniform: range/code, the rep distances and every branch live in scalar
// registers, the 11-bit probabilities (up to 14,135 of them, 28 KiB) live in
// LDS and come back through v_readfirstlane; the compressed input is read with
// scalar dword loads.  All 64 lanes take part in match copies, stored chunks
// and flushes.  Output is staged in an 8 KiB LDS ring (index == HBM address mod
// 16) that serves as the near dictionary; older history is read back from HBM
// through L2.  The block's own output span is the dictionary: every xz block
// starts with a dictionary reset (lzma2.lisp:97-98).
//
// bcj_x86_kernel / bcj_simple_kernel / delta_kernel: the block's filter chain, applied in place in
// HBM after LZMA2, then checksums.hip computes and compares the block check.
#include "semz_host.h"

namespace {

// Two sizes of the kernel's LDS state.  SMALL (4 KiB ring, literal table for lc + lp <= 3, which covers what xz
// writes by default: lc 3, lp 0) is 20,080 bytes, so eight units fit a CU, two per SIMD, and a SIMD with two waves
// issues twice the instructions of one with a lone wave.  A unit whose properties need the full table leaves the
// small kernel with LZ_RETRY_FULL and is decoded by the full-size kernel launched right behind it.
template <bool SMALL> struct LzSize {
    static constexpr int RING = SMALL ? 4096 : 8192;
    static constexpr int MASK = RING - 1;
    static constexpr int LCLP = SMALL ? 3 : 4;
    static constexpr int TOTAL = 1847 + (0x300 << LCLP) + 1;
};
constexpr int32_t LZ_RETRY_FULL = -101;
constexpr int LZ_FLUSH = 1024;                 // 64 lanes x 16 bytes
constexpr int MAX_MATCH = 273;                 // +lzma-max-expansion-length+, lzma.lisp:233-238

// probability layout (uint16 indices), lzma.lisp:220-231, 295-313, 428-460
enum {
    P_IS_LIT = 0, P_IS_SIMPLE = 192, P_USES_REP0 = 204, P_IS_SHORT = 216, P_IS_REP1 = 408, P_IS_REP2 = 420,
    P_LEN_SIMPLE = 432, P_LEN_REP = 946, P_SLOT = 1460, P_MID = 1716, P_ALIGN = 1831, P_LIT = 1847
};
enum { LEN_CHOICE = 0, LEN_LOW = 2, LEN_MID = 130, LEN_HIGH = 258 };

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

template <bool SMALL> struct __align__(16) LzLds {
    uint8_t ring[LzSize<SMALL>::RING];
    uint16_t probs[LzSize<SMALL>::TOTAL];
};

// Wave-uniform decoder state.  Reads go through next_byte(); after the first
// failure every later read returns 0 and the symbol loops run on until they
// poll `status`, which keeps the first failure (the one the reference raises).
// The hot state is small on purpose (every value that a rare branch changes
// costs a register copy on the common path): a 32-bit offset into a window of
// the input that is re-based at every chunk, one limit, the current dword.
constexpr int32_t LZ_PENDING_LIMIT = -100;     // a read ran into `end`: which limit it was is worked out afterwards

struct Lz {
    // byte source with nested limits (io.lisp:258-323); positions are v-offsets = offset inside `in` + (in & 3)
    const uint8_t *in0;   // in - (in & 3): dword aligned
    uint64_t vbase;       // v-offset of the window (multiple of 4)
    k_u32p base4;         // scalar-load view of in0 + vbase
    uint32_t off;         // next byte, relative to vbase
    uint32_t end;         // min of the three limits, relative to vbase (saturated: see rebase())
    uint32_t cur;         // unread bytes of the dword that holds byte `off`
    uint64_t phys_end;    // physical end of the input           -> SEMZ_EOF
    uint64_t block_end;   // declared xz compressed size, or ~0  -> SEMZ_E_XZ_BLOCK_OVERRUN
    uint64_t chunk_end;   // LZMA2 chunk compressed size, or ~0  -> SEMZ_E_LZMA2_CHUNK_OVERRUN
    int32_t status;
    uint32_t detail;
    // range decoder
    uint32_t range, code;

    __device__ __forceinline__ uint64_t pos() const { return vbase + off; }
    // written as selects: no control flow, so nothing else is merged behind it
    __device__ __forceinline__ void fail(int st, uint32_t d)
    {
        detail = status == SEMZ_OK ? d : detail;
        status = status == SEMZ_OK ? st : status;
    }
    // cur holds the unread bytes of the dword at `off`, already shifted down
    __device__ __forceinline__ void load_cur()
    {
        cur = 0;
        if (off < end) cur = base4[off >> 2] >> (8 * (off & 3));
    }
    // moves the window to `p` and recomputes the limit.  The window is re-based at every LZMA2 chunk (at most
    // 64 KiB long) and, for raw LZMA, every 2^30 bytes, so a saturated `end` is never reached
    __device__ __forceinline__ void rebase(uint64_t p)
    {
        vbase = p & ~3ull;
        base4 = (k_u32p)(uintptr_t)(in0 + vbase);
        off = (uint32_t)(p - vbase);
        uint64_t lim = phys_end < block_end ? phys_end : block_end;
        if (chunk_end < lim) lim = chunk_end;
        const uint64_t rel = lim > vbase ? lim - vbase : 0;
        end = rel < 0xFFFFFFF0ull ? (uint32_t)rel : 0xFFFFFFF0u;
        load_cur();
    }
    __device__ __forceinline__ void set_limits() { rebase(pos()); }
    // the failure of a read that ran into `end`: the innermost limit that has been reached wins; otherwise the input itself ended
    __device__ __forceinline__ void resolve_limit()
    {
        if (status != LZ_PENDING_LIMIT) return;
        const uint64_t p = pos();
        status = p >= chunk_end ? SEMZ_E_LZMA2_CHUNK_OVERRUN : p >= block_end ? SEMZ_E_XZ_BLOCK_OVERRUN : SEMZ_EOF;
        detail = 0;
    }
    __device__ __forceinline__ uint32_t next_byte()
    {
        if (off >= end) {
            fail(LZ_PENDING_LIMIT, 0);
            return 0;
        }
        const uint32_t b = cur & 0xFF;
        cur >>= 8;
        off++;
        if ((off & 3) == 0 && off < end) cur = base4[off >> 2];
        return b;
    }
    __device__ __forceinline__ void normalize()                          // lzma.lisp:64-73
    {
        if (range < (1u << 24)) {
            range <<= 8;
            code = (code << 8) | next_byte();
            if (!(code < range)) fail(SEMZ_E_LZMA_CODE_RANGE, 0);
        }
    }
    __device__ __forceinline__ uint32_t bit(uint16_t *probs, uint32_t idx)   // lzma.lisp:83-116
    {
        // selects instead of branches, and the update stored by every lane (same address, same value: one LDS
        // pass) instead of by lane 0 behind an exec mask: both halve the instructions of the step
        const uint32_t raw = probs[idx];                 // ds_read_u16 zero-extends
        uint32_t np, one;
        step<true>(raw, np, one);
        probs[idx] = (uint16_t)np;
        normalize();
        return one;
    }
    // The arithmetic of one bit, fourteen scalar instructions.  Written out: given the C form the compiler keeps
    // the comparison as a lane mask and takes the bit's value through a VGPR and back (five more instructions).
    // FROM_LANES: `pin` is the probability as the lanes loaded it (the same in every lane); it comes over with
    // v_readfirstlane inside the sequence, where the compiler would add a mask and a wait state around it.
    template <bool FROM_LANES> __device__ __forceinline__ void step(uint32_t pin, uint32_t &np, uint32_t &one)
    {
        uint32_t bound, t0, t1, t2, t3, p;
#define LZ_STEP_BODY                                                                                         \
            "s_lshr_b32 %[bound], %[range], 11\n\t"                                                           \
            "s_mul_i32 %[bound], %[bound], %[p]\n\t"                                                          \
            "s_lshr_b32 %[t0], %[p], 5\n\t"                                                                   \
            "s_sub_u32 %[t0], %[p], %[t0]\n\t"              /* p - (p >> 5) */                                \
            "s_sub_u32 %[t1], 0x800, %[p]\n\t"                                                                \
            "s_lshr_b32 %[t1], %[t1], 5\n\t"                                                                  \
            "s_add_u32 %[t1], %[p], %[t1]\n\t"              /* p + ((2048 - p) >> 5) */                       \
            "s_sub_u32 %[t2], %[range], %[bound]\n\t"                                                         \
            "s_sub_u32 %[t3], %[code], %[bound]\n\t"                                                          \
            "s_cmp_ge_u32 %[code], %[bound]\n\t"                                                              \
            "s_cselect_b32 %[range], %[t2], %[bound]\n\t"                                                     \
            "s_cselect_b32 %[code], %[t3], %[code]\n\t"                                                       \
            "s_cselect_b32 %[np], %[t0], %[t1]\n\t"                                                           \
            "s_cselect_b32 %[one], 1, 0"
        if constexpr (FROM_LANES) {
            asm("v_readfirstlane_b32 %[p], %[pv]\n\t" LZ_STEP_BODY
                : [range] "+s"(range), [code] "+s"(code), [np] "=&s"(np), [one] "=&s"(one), [bound] "=&s"(bound),
                  [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [p] "=&s"(p)
                : [pv] "v"(pin)
                : "scc");
        } else {
            p = pin;
            asm(LZ_STEP_BODY
                : [range] "+s"(range), [code] "+s"(code), [np] "=&s"(np), [one] "=&s"(one), [bound] "=&s"(bound),
                  [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3)
                : [p] "s"(p)
                : "scc");
        }
#undef LZ_STEP_BODY
    }
    // same over probabilities kept in HBM (the literal coder of SEMZ_UNIT_LZMA_WIDE); L2-coherent accesses, so
    // that the wave reads back its own updates
    __device__ __forceinline__ uint32_t bit_g(uint16_t *probs, uint32_t idx)
    {
        const uint32_t p = uni((uint32_t)__hip_atomic_load(probs + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        uint32_t np, one;
        step<false>(p, np, one);
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(probs + idx, (uint16_t)np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        normalize();
        return one;
    }
    __device__ __forceinline__ uint32_t be_bits(uint16_t *probs, uint32_t base, uint32_t count)   // lzma.lisp:122-135
    {
        const uint32_t bound = 1u << count;
        uint32_t m = 1;
        while (m < bound) m = (m << 1) | bit(probs, base + m);
        return m - bound;
    }
    __device__ __forceinline__ uint32_t le_bits(uint16_t *probs, uint32_t base, uint32_t count)   // lzma.lisp:138-145
    {
        return __brev(be_bits(probs, base, count)) >> (32 - count);
    }
    __device__ __forceinline__ uint32_t fixed_bits(uint32_t count)                            // lzma.lisp:174-192
    {
        uint32_t res = 0;
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t bound = range >> 1;
            range = bound;
            if (code >= bound) {
                code -= bound;
                if (code == bound) fail(SEMZ_E_LZMA_CODE_RANGE, 0);
                res = (res << 1) | 1;
            } else res <<= 1;
            normalize();
        }
        return res;
    }
    __device__ __forceinline__ uint32_t length(uint16_t *probs, uint32_t dec, uint32_t pos_state)   // lzma.lisp:247-257
    {
        if (bit(probs, dec + LEN_CHOICE) == 0) return be_bits(probs, dec + LEN_LOW + 8 * pos_state, 3);
        if (bit(probs, dec + LEN_CHOICE + 1) == 0) return 8 + be_bits(probs, dec + LEN_MID + 8 * pos_state, 3);
        return 16 + be_bits(probs, dec + LEN_HIGH, 8);
    }
    __device__ __forceinline__ uint32_t distance(uint16_t *probs, uint32_t raw_len)            // lzma.lisp:323-341
    {
        const uint32_t slot = be_bits(probs, P_SLOT + 64 * (raw_len < 3 ? raw_len : 3), 6);
        if (slot < 4) return slot;
        const uint32_t bit_count = (slot >> 1) - 1;
        const uint32_t min_dist = (2u | (slot & 1)) << bit_count;
        if (slot < 14) return min_dist + le_bits(probs, P_MID + min_dist - slot, bit_count);
        const uint32_t hi = fixed_bits(bit_count - 4) << 4;
        return min_dist + hi + le_bits(probs, P_ALIGN, 4);
    }
};

__device__ __forceinline__ uint32_t ld_bytes_le(const uint8_t *p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

enum { EOF_MAYBE = 0, EOF_ALWAYS = 1, EOF_NEVER = 2 };          // eof-mode, lzma.lisp:575-577
constexpr uint32_t WIDE_LIT_PROBS = 0x300u << 12;               // lc <= 8, lp <= 4 (lzma.lisp:818-819)

// Units the accept-only kernel of lzma2_fast.hip handed to this one, since the library was loaded (option
// "lzma-handed-back": tests make sure that well-formed blocks never take the slow road).
__device__ unsigned g_lz_handed_back;

// WIDE: the literal coder's probabilities live in `wide` (WIDE_LIT_PROBS entries per unit) instead of LDS
// SMALL: see LzSize; retry_pass: only the units that the small kernel left with LZ_RETRY_FULL are decoded
template <bool WIDE, bool SMALL>
__global__ __launch_bounds__(64, 1) void lzma2_kernel(const uint8_t *in, uint64_t in_len, const semz_unit *units, uint32_t nunits,
                                                      uint8_t *out, uint64_t out_total, semz_result *results, semz_tail *tails,
                                                      uint16_t *wide, int retry_pass)
{
    constexpr int LZ_RING = LzSize<SMALL>::RING, LZ_MASK = LzSize<SMALL>::MASK, P_TOTAL = LzSize<SMALL>::TOTAL;
    constexpr uint32_t MAX_LCLP = LzSize<SMALL>::LCLP;
    __shared__ LzLds<SMALL> L;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    if (retry_pass && uni(__hip_atomic_load(&results[u].status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != LZ_RETRY_FULL) return;
    if (SMALL && retry_pass == 2 && lane == 0) atomicAdd(&g_lz_handed_back, 1u);
    semz_unit un;
    {
        k_u32p up = (k_u32p)(uintptr_t)(units + u);
        uint32_t w[sizeof(semz_unit) / 4];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(semz_unit) / 4); i++) w[i] = up[i];
        __builtin_memcpy(&un, w, sizeof un);
    }
    if (!semz_unit_spans_ok(un, in_len, out_total, 2)) {         // nothing of this unit is touched
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
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);
    const uint64_t out_cap = un.out_cap;
    const uint32_t dict_size = un.param0;
    const bool is_xz = un.kind == SEMZ_UNIT_XZ_BLOCK;
    const bool is_lzma1 = un.kind == SEMZ_UNIT_LZMA || un.kind == SEMZ_UNIT_LZMA_WIDE;
    uint16_t *const wlit = WIDE ? wide + (size_t)u * WIDE_LIT_PROBS : nullptr;
    uint8_t *safe_byte = out;                          // always-readable address for masked-off gather lanes

    const uint64_t amis = (uint64_t)((uintptr_t)in & 3);
    Lz z;
    z.in0 = in - amis;
    const uint64_t pos0 = un.in_off + amis;
    z.phys_end = un.in_limit + amis;
    z.block_end = (is_xz && (un.flags & SEMZ_UF_HAVE_IN_SIZE)) ? pos0 + un.size_in : ~0ull;
    if (z.block_end < pos0) z.block_end = ~0ull;       // size_in near 2^64: cannot be reached
    z.chunk_end = ~0ull;
    z.status = SEMZ_OK;
    z.detail = 0;
    z.range = 0; z.code = 0;
    z.rebase(pos0);
    const uint64_t data_start = pos0;

    // LZMA state variables (lzma-state-vars, lzma.lisp:428-460)
    uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0, history = 0, total_i = 0;
    uint32_t lc = 0, lp_mask = 0, pb_mask = 0;
    bool have_state = false;
    enum { NEED_DICT = 0, NEED_PROPS = 1, NEED_NONE = 2 };
    int need = NEED_DICT;

    uint64_t p = 0;            // output position
    uint64_t dict_base = 0;    // position of the last dictionary reset
    uint64_t flushed = 0;

    // moves whole 1 KiB pieces of the ring to HBM; `all` also moves the tail
    auto flush = [&](bool all) {
        const uint32_t head = (16 - shift) & 15;
        if (flushed == 0 && head && (p >= head || all)) {
            const uint32_t n = p < head ? (uint32_t)p : head;
            if (lane < n) ob[lane] = L.ring[(lane + shift) & LZ_MASK];
            flushed = n;
        }
        flushed = uni(flushed);         // see the note at the end
        if (flushed || !head) {
            while (p - flushed >= (uint64_t)LZ_FLUSH) {
                const uint64_t a = flushed + 16u * lane;
                const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & LZ_MASK]);
                *reinterpret_cast<uint4 *>(ob + a) = v;
                flushed += LZ_FLUSH;
            }
            if (all) {
                const uint32_t pieces = (uint32_t)((p - flushed) >> 4);
                if (lane < pieces) {
                    const uint64_t a = flushed + 16u * lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(uint32_t)(a + shift) & LZ_MASK]);
                    *reinterpret_cast<uint4 *>(ob + a) = v;
                }
                flushed += 16ull * pieces;
                const uint32_t rest = (uint32_t)(p - flushed);
                if (lane < rest) ob[flushed + lane] = L.ring[(uint32_t)(flushed + lane + shift) & LZ_MASK];
                flushed = p;
            }
        }
        // the optimiser threads the lane-masked head store into the update of `flushed`, which then counts as
        // divergent and drags the whole decoder state into VGPRs; this pins it as the wave-uniform value it is
        flushed = uni(flushed);
    };
    // byte of history at position s < p: the ring holds [p - (LZ_RING - MAX_MATCH - LZ_FLUSH) .. p) at least
    auto history_byte = [&](uint64_t s) -> uint32_t {
        if (s + (LZ_RING - MAX_MATCH) > p) return uni((uint32_t)L.ring[(uint32_t)(s + shift) & LZ_MASK]);
        return uni((uint32_t)__hip_atomic_load(ob + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };

    // literal coder bit: probabilities in LDS, or in HBM for the wide contexts of :raw-lzma
    auto lit_bit = [&](uint32_t idx) -> uint32_t {
        if constexpr (WIDE) return z.bit_g(wlit, idx);
        else return z.bit(L.probs, P_LIT + idx);
    };

    // ---- LZMA2 control loop (lzma2.lisp:77-204); a :raw-lzma unit passes through it once as one "chunk" ----
    bool lzma1_started = false;
    for (;;) {
        if (z.status != SEMZ_OK) break;
        uint64_t end;                  // the variable P of lzma.lisp:532-551; ~0 = no declared size
        uint64_t cap_end = ~0ull;      // where the output span ends, for data without a declared size
        uint32_t eof_mode = EOF_NEVER; // lzma2.lisp:133
        bool fresh;
        if (is_lzma1) {
            // :raw-lzma (lzma.lisp:809-835): properties and sizes come from the caller or the .lzma header
            if (lzma1_started) break;
            lzma1_started = true;
            const uint32_t nlc = un.param1 & 0xFF, nlp = (un.param1 >> 8) & 0xFF, npb = (un.param1 >> 16) & 0xFF;
            if (nlc > 8 || nlp > 4 || npb > 4 || (!WIDE && nlc + nlp > 4) || un.param2 > EOF_NEVER) {
                z.fail(SEMZ_E_UNSUPPORTED, 2);        // check-type in the reference; wide contexts need SEMZ_UNIT_LZMA_WIDE
                break;
            }
            if (!WIDE && nlc + nlp > MAX_LCLP) { z.fail(LZ_RETRY_FULL, 0); break; }
            lc = nlc; lp_mask = (1u << nlp) - 1; pb_mask = (1u << npb) - 1;
            end = (un.flags & SEMZ_UF_HAVE_OUT_SIZE) ? un.size_out : ~0ull;
            cap_end = out_cap;
            eof_mode = un.param2;
            fresh = true;
        } else {
            const uint32_t control = z.next_byte();
            if (z.status != SEMZ_OK) break;
            if (control == 0) break;                                 // end of the LZMA2 stream
            if (control < 0x80) {
                if (control > 2) { z.fail(SEMZ_E_LZMA2_CONTROL, control); break; }
                // stored chunk (lzma2.lisp:152-189)
                const uint32_t b0 = z.next_byte(), b1 = z.next_byte();
                if (z.status != SEMZ_OK) break;
                const uint64_t usize = ((b0 << 8) | b1) + 1;
                if (control == 2) {
                    if (need == NEED_DICT) { z.fail(SEMZ_E_LZMA2_NEED_DICT_RESET, 0); break; }
                    need = NEED_NONE;
                } else {
                    need = NEED_PROPS;
                    dict_base = p;
                }
                if (p + usize > out_cap) { z.fail(SEMZ_E_OUT_CAP, 0); break; }
                if (usize > (uint64_t)(z.end - z.off)) {                 // bs-read-sequence with eof-error-p
                    z.off = z.end;
                    (void)z.next_byte();
                    break;
                }
                const uint8_t *src = in + (z.pos() - amis);
                lds_sync();
                for (uint64_t off = 0; off < usize; off += LZ_FLUSH) {
                    const uint32_t n = usize - off < (uint64_t)LZ_FLUSH ? (uint32_t)(usize - off) : (uint32_t)LZ_FLUSH;
                    for (uint32_t k = lane; k < n; k += 64) L.ring[(uint32_t)(p + k + shift) & LZ_MASK] = src[off + k];
                    p += n;
                    lds_sync();
                    flush(false);
                }
                z.rebase(z.pos() + usize);
                if (have_state) total_i = (total_i + (uint32_t)usize) & 15;   // lzma2.lisp:184-188
                continue;
            }
            uint32_t b0 = z.next_byte(), b1 = z.next_byte(), b2 = z.next_byte(), b3 = z.next_byte();
            if (z.status != SEMZ_OK) break;
            const uint64_t usize = (((uint64_t)(control & 31) << 16) | (b0 << 8) | b1) + 1;
            const uint64_t csize = ((b2 << 8) | b3) + 1;
            if (need == NEED_DICT && control < 0xE0) { z.fail(SEMZ_E_LZMA2_NEED_DICT_RESET, 0); break; }
            if (need == NEED_PROPS && control < 0xC0) { z.fail(SEMZ_E_LZMA2_NEED_PROPS, 0); break; }
            need = NEED_NONE;
            if (control >= 0xC0) {
                const uint32_t octet = z.next_byte();               // parse-lzma2-props, lzma2.lisp:31-38
                if (z.status != SEMZ_OK) break;
                const uint32_t nlc = octet % 9, nlp = (octet / 9) % 5, npb = octet / 45;
                if (npb > 4) { z.fail(SEMZ_E_LZMA_PB, npb); break; }
                if (nlc + nlp > 4) { z.fail(SEMZ_E_LZMA2_LCLP, nlc + nlp); break; }
                if (!WIDE && nlc + nlp > MAX_LCLP) { z.fail(LZ_RETRY_FULL, 0); break; }
                lc = nlc; lp_mask = (1u << nlp) - 1; pb_mask = (1u << npb) - 1;
            } else if (!have_state) { z.fail(SEMZ_E_LZMA2_NO_STATE, 0); break; }
            if (control >= 0xE0) dict_base = p;
            if (p + usize > out_cap) { z.fail(SEMZ_E_OUT_CAP, 0); break; }
            // counted source for the chunk (lzma2.lisp:111-116)
            z.chunk_end = z.pos() + csize;
            z.set_limits();
            end = p + usize;
            fresh = control >= 0xA0 || !have_state;                  // fresh lzma-state-vars (lzma2.lisp:144-145)
        }
        // the range decoder (lzma.lisp:55-62)
        if (z.next_byte() != 0) z.fail(SEMZ_E_LZMA_FIRST_BYTE, 0);
        z.range = 0xFFFFFFFFu;
        z.code = 0;
        for (int i = 0; i < 4; i++) z.code = (z.code << 8) | z.next_byte();
        if (z.status == SEMZ_OK && !(z.code < z.range)) z.fail(SEMZ_E_LZMA_CODE_START, 0);
        if (z.status != SEMZ_OK) break;
        if (fresh) {
            for (uint32_t i = lane; i < (uint32_t)P_TOTAL; i += 64) L.probs[i] = 1024;
            if constexpr (WIDE) {
                const uint32_t n8 = (0x300u << (lc + __builtin_popcount(lp_mask))) / 8;      // 16-byte pieces
                const uint4 fill = make_uint4(0x04000400u, 0x04000400u, 0x04000400u, 0x04000400u);
                for (uint32_t i = lane; i < n8; i += 64) reinterpret_cast<uint4 *>(wlit)[i] = fill;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            }
            rep0 = rep1 = rep2 = rep3 = 0; history = 0; total_i = 0;
            lds_sync();
        }
        have_state = true;

        // ---- decode-lzma (lzma.lisp:566-775) ----
        bool eofp = false;
        for (;;) {
            if (z.status != SEMZ_OK) break;
            if (z.off >= 0x40000000u) z.set_limits();                // raw LZMA has no chunks: re-base the input window
            const uint32_t pos_state = total_i & pb_mask;
            if (p == end) {
                // handle-eof-insanity (lzma.lisp:640-659): :always probes at once, the others ask the range decoder first
                if (eof_mode == EOF_ALWAYS || z.code != 0) {
                    bool marker = false;                             // probe-eof, lzma.lisp:623-638
                    if (z.bit(L.probs, P_IS_LIT + (history << 4) + pos_state) == 1 &&
                        z.bit(L.probs, P_IS_SIMPLE + history) == 0) {
                        const uint32_t len = z.length(L.probs, P_LEN_SIMPLE, pos_state);
                        marker = z.distance(L.probs, len) == 0xFFFFFFFFu;
                    }
                    if (eof_mode == EOF_NEVER) z.fail(marker ? SEMZ_E_LZMA_ILLEGAL_EOF : SEMZ_E_LZMA_CONTINUES, 0);
                    else if (!marker) z.fail(eof_mode == EOF_ALWAYS ? SEMZ_E_LZMA_NO_EOF_MARKER : SEMZ_E_LZMA_NO_END, 0);
                }
                eofp = true;
                break;
            }
            uint32_t len;
            if (z.bit(L.probs, P_IS_LIT + (history << 4) + pos_state) == 0) {
                // decode-literal, lzma.lisp:661-688
                const uint32_t prev = p == dict_base ? 0u : uni((uint32_t)L.ring[(uint32_t)(p - 1 + shift) & LZ_MASK]);
                const uint32_t base = 0x300u * (((total_i & lp_mask) << lc) | (prev >> (8 - lc)));
                uint32_t symbol = 1;
                if (history >= 7) {
                    // decode-lzma-literal/match-case, lzma.lisp:497-518
                    if (p - dict_base < (uint64_t)rep0 + 1) { z.fail(SEMZ_E_UNSUPPORTED, 1); break; }
                    const uint32_t match_byte = history_byte(p - 1 - rep0);
                    for (int bit_i = 7; bit_i >= 0; bit_i--) {
                        const uint32_t mb = (match_byte >> bit_i) & 1;
                        const uint32_t b = lit_bit(base + 0x100 + (mb << 8) + symbol);
                        symbol = (symbol << 1) | b;
                        if (mb != b) break;
                    }
                }
                while (symbol < 0x100) symbol = (symbol << 1) | lit_bit(base + symbol);
                if (p == cap_end) { z.fail(SEMZ_E_OUT_CAP, 0); break; }
                L.ring[(uint32_t)(p + shift) & LZ_MASK] = (uint8_t)symbol;      // every lane: same address, same value
                p++;
                total_i++;
                history = history < 4 ? 0 : history < 10 ? history - 3 : history - 6;   // lzma.lisp:615
                if ((uint32_t)p - (uint32_t)flushed >= (uint32_t)LZ_FLUSH + 16) { lds_sync(); flush(false); }   // (the difference is below the ring size)
                continue;
            }
            if (z.bit(L.probs, P_IS_SIMPLE + history) == 0) {
                const uint32_t rl = z.length(L.probs, P_LEN_SIMPLE, pos_state);
                history = history < 7 ? 7 : 10;
                rep3 = rep2; rep2 = rep1; rep1 = rep0;
                rep0 = z.distance(L.probs, rl);
                if (rep0 == 0xFFFFFFFFu) {                                                  // lzma.lisp:727-732
                    if (eof_mode == EOF_NEVER) z.fail(SEMZ_E_LZMA_ILLEGAL_EOF, 0);
                    else eofp = true;
                    break;
                }
                len = rl + 2;
            } else if (z.bit(L.probs, P_USES_REP0 + history) == 0) {
                if (z.bit(L.probs, P_IS_SHORT + (history << 4) + pos_state) == 0) {
                    history = history < 7 ? 9 : 11;
                    len = 1;
                } else {
                    history = history < 7 ? 8 : 11;
                    len = 2 + z.length(L.probs, P_LEN_REP, pos_state);
                }
            } else {
                uint32_t t;
                if (z.bit(L.probs, P_IS_REP1 + history) == 0) { t = rep1; rep1 = rep0; rep0 = t; }
                else if (z.bit(L.probs, P_IS_REP2 + history) == 0) { t = rep2; rep2 = rep1; rep1 = rep0; rep0 = t; }
                else { t = rep3; rep3 = rep2; rep2 = rep1; rep1 = rep0; rep0 = t; }
                history = history < 7 ? 8 : 11;
                len = 2 + z.length(L.probs, P_LEN_REP, pos_state);
            }
            if (z.status != SEMZ_OK) break;
            // handle-match, lzma.lisp:600-612
            const uint64_t dist = (uint64_t)rep0 + 1;
            if (!(dist <= dict_size)) { z.fail(SEMZ_E_LZMA_DICT_SIZE, 0); break; }
            if (!(dist <= p - dict_base)) { z.fail(SEMZ_E_LZMA_NO_DATA, 0); break; }
            if (!(len <= end - p)) { z.fail(SEMZ_E_LZMA_MATCH_SIZE, 0); break; }
            if (!(len <= cap_end - p)) { z.fail(SEMZ_E_OUT_CAP, 0); break; }
            {
                // copy-match (common.lisp:324-333), all lanes; sources older than the ring come from HBM
                lds_sync();
                const uint64_t src = p - dist;
                const bool all_near = src + (LZ_RING - MAX_MATCH) > p + len;
                for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    const uint64_t s = src + (dist >= len ? k : k % (uint32_t)dist);
                    uint32_t b;
                    if (all_near) b = L.ring[(uint32_t)(s + shift) & LZ_MASK];
                    else {
                        const bool useg = k < len && s + (LZ_RING - MAX_MATCH) <= p;
                        const uint32_t gv = __hip_atomic_load(useg ? ob + s : safe_byte, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t lv = L.ring[(uint32_t)(s + shift) & LZ_MASK];
                        b = useg ? gv : lv;
                    }
                    if (k < len) L.ring[(uint32_t)(p + k + shift) & LZ_MASK] = (uint8_t)b;
                }
                lds_sync();
            }
            p += len;
            total_i += len;
            if ((uint32_t)p - (uint32_t)flushed >= (uint32_t)LZ_FLUSH + 16) flush(false);
        }
        total_i &= 15;
        if (z.status != SEMZ_OK) break;
        if (eofp && z.code != 0) { z.fail(SEMZ_E_LZMA_FINISH_STATE, 0); break; }          // lzma.lisp:772-774
        if (is_lzma1) {
            if (end != ~0ull && p != end) z.fail(SEMZ_E_LZMA_EARLY_END, 0);                 // lzma.lisp:891-894
            break;
        }
        if (z.pos() != z.chunk_end) { z.fail(SEMZ_E_LZMA2_CHUNK_SHORT, 0); break; }   // lzma2.lisp:200-201
        z.chunk_end = ~0ull;
        z.set_limits();
    }
    lds_sync();
    flush(true);

    // ---- xz block end (xz.lisp:426-457) and results ----
    semz_tail t{};
    semz_result r{};
    z.resolve_limit();
    const uint64_t real_csize = z.pos() - data_start;
    if (is_xz && z.status == SEMZ_OK) {
        if ((un.flags & SEMZ_UF_HAVE_IN_SIZE) && real_csize != un.size_in) z.fail(SEMZ_E_XZ_BLOCK_SHORT, 0);
        else if ((un.flags & SEMZ_UF_HAVE_OUT_SIZE) && p != un.size_out) z.fail(SEMZ_E_XZ_BLOCK_OUT_SIZE, 0);
        else {
            z.block_end = ~0ull;
            z.set_limits();
            const uint32_t pad = (4 - ((uint32_t)real_csize & 3)) & 3;
            for (uint32_t k = 0; k < pad && z.status == SEMZ_OK; k++)
                if (z.next_byte() != 0) z.fail(SEMZ_E_XZ_BLOCK_PADDING, 0);
            if (z.status == SEMZ_OK) {
                const uint32_t ctype = un.param1;
                const uint32_t cs = ctype == SEMZ_CHECK_CRC32 ? 4 : ctype == SEMZ_CHECK_CRC64 ? 8 : ctype == SEMZ_CHECK_SHA256 ? 32 : 0;
                t.check_kind = ctype == SEMZ_CHECK_CRC32 ? SEMZ_TK_CRC32 : ctype == SEMZ_CHECK_CRC64 ? SEMZ_TK_CRC64
                               : ctype == SEMZ_CHECK_SHA256 ? SEMZ_TK_SHA256 : SEMZ_TK_NONE;
                uint64_t endpos = z.pos();
                if (endpos + cs <= z.phys_end) {
                    t.have_check = 1;
                    const uint8_t *cp = in + (endpos - amis);
                    for (uint32_t k = 0; k < cs / 4; k++) t.want[k] = ld_bytes_le(cp + 4 * k, 4);
                    endpos += cs;
                } else {
                    endpos = z.phys_end;                             // bs-read-le runs into the end: eof
                }
                z.rebase(endpos);
            }
        }
    }
    z.resolve_limit();
    t.trailer_end_bit = 8 * (z.pos() - amis);
    r.status = z.status;
    r.detail = z.status == SEMZ_OK ? (uint32_t)(real_csize >> 32) : z.detail;
    r.aux = (uint32_t)real_csize;
    r.in_end_bit = 8 * (z.pos() - amis);
    r.out_len = p;
    r.out_off = un.out_off;
    if (lane == 0) {
        results[u] = r;
        tails[u] = t;
    }
}

// ---- x86 BCJ inverse in place (xz-bcj-filters.lisp:169-219), one wave per unit ----
// The filter state changes only at E8/E9 bytes.  The block is walked in 1 KiB
// pieces staged in LDS: all lanes mark the candidate bytes of the piece, then
// the wave-uniform state machine visits the candidates in order and patches
// the operands in LDS; the piece is written back with 16-byte stores.  A
// 4-byte overlap carries operands that straddle pieces.
constexpr int BCJ_PIECE = 1024;

__global__ __launch_bounds__(64) void bcj_x86_kernel(const semz_unit *units, uint32_t nunits, uint8_t *out,
                                                     const semz_result *results, int stage)
{
    __shared__ __align__(16) uint8_t buf[BCJ_PIECE + 16];
    __shared__ uint32_t cand[BCJ_PIECE / 32];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const semz_unit un = units[u];
    const semz_result r = results[u];
    if (r.status != SEMZ_OK || ((un.filters >> (4 * stage)) & 15) != SEMZ_FILTER_X86) return;
    uint8_t *ob = out + r.out_off;
    const uint64_t n = r.out_len;
    if (n < 5) return;
    const uint64_t limit = n - 5;                      // candidates at i <= limit are processed
    const uint32_t now_pos0 = un.param2;
    uint32_t prev_mask = 0, prev_pos = 0u - 5u;
    if ((uint32_t)(now_pos0 - prev_pos) > 5) prev_pos = now_pos0 - 5;
    uint64_t next_i = 0;                               // first byte not yet consumed by the state machine
    if (((uintptr_t)ob & 15) == 0) {
        // The usual block starts on a 16-byte boundary: a piece is one 16-byte load and one 16-byte store a lane, the
        // candidate bits come out of the registers (a byte is E8 or E9 when (b ^ E8) & FE is zero: the zero-byte test
        // on four bytes at a time, squeezed to four bits by a multiplication), the next piece is on its way while
        // this one's candidates are visited, and the four bytes behind a piece, which an operand that straddles
        // the boundary patches, are handed on in LDS instead of through memory.  (Byte loads, a byte-wise scan and
        // no load in flight made this kernel 8 ms of the xz step.)
        auto load_piece = [&](uint64_t base) -> uint4 {
            const uint64_t at = base + 16 * lane;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (at + 16 <= n) x = *reinterpret_cast<const uint4 *>(ob + at);
            else if (at < n) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < 16 && at + k < n; k++) w[k >> 2] |= (uint32_t)ob[at + k] << (8 * (k & 3));
                x = make_uint4(w[0], w[1], w[2], w[3]);
            }
            return x;
        };
        uint4 cur = load_piece(0);
        uint32_t carry = 0;                                 // the piece's first four bytes as the piece before left them
        for (uint64_t base = 0; base <= limit; base += BCJ_PIECE) {
            if (base) cur.x = lane == 0 ? carry : cur.x;
            *reinterpret_cast<uint4 *>(buf + 16 * lane) = cur;
            // the four bytes behind the piece and the next piece, while this one is worked on
            const uint4 nxt = load_piece(base + BCJ_PIECE);
            if (lane == 0) *reinterpret_cast<uint32_t *>(buf + BCJ_PIECE) = nxt.x;
            {
                const uint32_t ws[4] = {cur.x, cur.y, cur.z, cur.w};
                uint32_t m = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t t = (ws[q] ^ 0xE8E8E8E8u) & 0xFEFEFEFEu;
                    const uint32_t z = (t - 0x01010101u) & ~t & 0x80808080u;      // 0x80 in every byte that is zero (no byte of t is 1)
                    m |= ((((z >> 7) * 0x00204081u) >> 21) & 15u) << (4 * q);
                }
                const uint64_t first = base + 16 * lane;
                const uint32_t valid = first > limit ? 0u : (limit - first >= 15 ? 16u : (uint32_t)(limit - first) + 1u);
                m &= (1u << valid) - 1;
                reinterpret_cast<uint16_t *>(cand)[lane] = (uint16_t)m;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // (the bitmap comes back as one register, lane w = word w, and only the words that hold a candidate are
            // visited; a candidate's five bytes come in one LDS round trip, two aligned words and a shift)
            const uint32_t candv = cand[lane & (BCJ_PIECE / 32 - 1)];
            uint32_t wordm = (uint32_t)__builtin_amdgcn_ballot_w64(lane < BCJ_PIECE / 32 && candv != 0);
            while (wordm) {
                const uint32_t w = (uint32_t)__builtin_ctz(wordm);
                wordm &= wordm - 1;
                uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)candv, (int)w);
                while (m) {
                    const uint32_t bitn = __builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t li = 32 * w + bitn;                  // index inside the piece
                    const uint64_t i = base + li;
                    if (i < next_i) continue;                            // inside an operand that was converted
                    const uint32_t offset = now_pos0 + (uint32_t)i - prev_pos;
                    prev_pos = now_pos0 + (uint32_t)i;
                    if (offset > 5) prev_mask = 0;
                    else for (uint32_t k = 0; k < offset; k++) prev_mask = (prev_mask & 0x77) << 1;
                    const uint32_t *bw = reinterpret_cast<const uint32_t *>(buf + ((li + 1) & ~3u));
                    const uint32_t opnd = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)(uint32_t)(((((uint64_t)bw[1] << 32) | bw[0]) >> (8 * ((li + 1) & 3)))));   // bytes li + 1 .. li + 4
                    uint32_t b = opnd >> 24;
                    const bool allowed = (0x17u >> ((prev_mask >> 1) & 7)) & 1;   // t t t nil t nil nil nil
                    if ((b == 0x00 || b == 0xFF) && allowed && (prev_mask >> 1) < 0x10) {
                        uint32_t src = opnd;
                        uint32_t dest;
                        for (;;) {
                            dest = src - (now_pos0 + (uint32_t)i + 5);
                            if (prev_mask == 0) break;
                            const uint32_t j = (0x33332210u >> (4 * (prev_mask >> 1))) & 15;   // 0 1 2 2 3 3 3 3
                            b = (dest >> (24 - 8 * j)) & 0xFF;
                            if (!(b == 0x00 || b == 0xFF)) break;
                            src = dest ^ ((1u << (32 - 8 * j)) - 1);
                        }
                        if (lane == 0) {
                            buf[li + 4] = (uint8_t)~(((dest >> 24) & 1) - 1);
                            buf[li + 1] = (uint8_t)dest;
                            buf[li + 2] = (uint8_t)(dest >> 8);
                            buf[li + 3] = (uint8_t)(dest >> 16);
                        }
                        next_i = i + 5;
                        prev_mask = 0;
                    } else {
                        next_i = i + 1;
                        prev_mask |= 1;
                        if (b == 0x00 || b == 0xFF) prev_mask |= 0x10;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // back to memory: the piece, and behind the block's last piece the bytes that piece patched
            {
                const uint4 y = *reinterpret_cast<const uint4 *>(buf + 16 * lane);
                const uint64_t at = base + 16 * lane;
                if (at + 16 <= n) *reinterpret_cast<uint4 *>(ob + at) = y;
                else if (at < n) {
                    const uint32_t ys[4] = {y.x, y.y, y.z, y.w};
                    for (uint32_t k = 0; k < 16 && at + k < n; k++) ob[at + k] = (uint8_t)(ys[k >> 2] >> (8 * (k & 3)));
                }
            }
            carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const uint32_t *>(buf + BCJ_PIECE));
            const bool more = base + BCJ_PIECE <= limit;
            if (!more && lane < 4 && base + BCJ_PIECE + lane < n) ob[base + BCJ_PIECE + lane] = (uint8_t)(carry >> (8 * lane));
            cur = nxt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        return;
    }
    for (uint64_t base = 0; base <= limit; base += BCJ_PIECE) {
        // stage [base, base + BCJ_PIECE + 4)
        const uint64_t span = n - base < (uint64_t)(BCJ_PIECE + 4) ? n - base : (uint64_t)(BCJ_PIECE + 4);
        for (uint32_t k = lane; k < span; k += 64) buf[k] = ob[base + k];
        if (lane < BCJ_PIECE / 32) cand[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // candidate bitmap: lane marks its 16 bytes
        {
            uint32_t m = 0;
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t i = 16 * lane + k;
                const uint8_t b = buf[i];
                if ((b == 0xE8 || b == 0xE9) && base + i <= limit && i < (uint32_t)BCJ_PIECE) m |= 1u << k;
            }
            if (m) atomicOr(&cand[lane >> 1], m << (16 * (lane & 1)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t w = 0; w < BCJ_PIECE / 32; w++) {
            uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)cand[w]);
            while (m) {
                const uint32_t bitn = __builtin_ctz(m);
                m &= m - 1;
                const uint32_t li = 32 * w + bitn;                  // index inside the piece
                const uint64_t i = base + li;
                if (i < next_i) continue;                            // inside an operand that was converted
                const uint32_t offset = now_pos0 + (uint32_t)i - prev_pos;
                prev_pos = now_pos0 + (uint32_t)i;
                if (offset > 5) prev_mask = 0;
                else for (uint32_t k = 0; k < offset; k++) prev_mask = (prev_mask & 0x77) << 1;
                uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)buf[li + 4]);
                const bool allowed = (0x17u >> ((prev_mask >> 1) & 7)) & 1;   // t t t nil t nil nil nil
                if ((b == 0x00 || b == 0xFF) && allowed && (prev_mask >> 1) < 0x10) {
                    uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)((uint32_t)buf[li + 1] | ((uint32_t)buf[li + 2] << 8) | ((uint32_t)buf[li + 3] << 16) | (b << 24)));
                    uint32_t dest;
                    for (;;) {
                        dest = src - (now_pos0 + (uint32_t)i + 5);
                        if (prev_mask == 0) break;
                        const uint32_t j = (0x33332210u >> (4 * (prev_mask >> 1))) & 15;   // 0 1 2 2 3 3 3 3
                        b = (dest >> (24 - 8 * j)) & 0xFF;
                        if (!(b == 0x00 || b == 0xFF)) break;
                        src = dest ^ ((1u << (32 - 8 * j)) - 1);
                    }
                    if (lane == 0) {
                        buf[li + 4] = (uint8_t)~(((dest >> 24) & 1) - 1);
                        buf[li + 1] = (uint8_t)dest;
                        buf[li + 2] = (uint8_t)(dest >> 8);
                        buf[li + 3] = (uint8_t)(dest >> 16);
                    }
                    next_i = i + 5;
                    prev_mask = 0;
                } else {
                    next_i = i + 1;
                    prev_mask |= 1;
                    if (b == 0x00 || b == 0xFF) prev_mask |= 0x10;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t k = lane; k < span; k += 64) ob[base + k] = buf[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// ---- delta filter inverse in place (xz.lisp:238-258) ----
// out[i] = in[i] + out[i - d] (mod 256): an inclusive prefix sum along each residue class mod d.
// One wave per unit; lane L owns the residues L, L + 64, ... < d and walks them serially.
__global__ __launch_bounds__(64) void delta_kernel(const semz_unit *units, uint32_t nunits, uint8_t *out,
                                                   const semz_result *results, int stage)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const semz_unit un = units[u];
    const semz_result r = results[u];
    if (r.status != SEMZ_OK || ((un.filters >> (4 * stage)) & 15) != SEMZ_FILTER_DELTA) return;
    uint8_t *ob = out + r.out_off;
    const uint64_t n = r.out_len;
    int delta_index = 0;                               // delta stages before this one
    for (int k = 0; k < stage; k++) delta_index += ((un.filters >> (4 * k)) & 15) == SEMZ_FILTER_DELTA;
    const uint32_t d = 1 + ((un.param3 >> (8 * delta_index)) & 0xFF);
    for (uint32_t res = lane; res < d; res += 64) {
        uint8_t acc = 0;
        for (uint64_t i = res; i < n; i += d) {
            acc = (uint8_t)(acc + ob[i]);
            ob[i] = acc;
        }
    }
}

// ---- the other BCJ inverses in place (xz-bcj-filters.lisp:45-167), one 256-thread workgroup per unit ----
// These filters look at fixed positions (every 4th byte, every 2nd for ARM-Thumb, every 16th for IA-64),
// counted from the start of the block, and no conversion changes the bytes another position tests: an
// ARM-Thumb BL pair at i has (b[i+1] & 0xF8) == 0xF0 and (b[i+3] & 0xF8) == 0xF8 before and after, which
// rules out a pair at i - 2 and at i + 2 (the reference's "skip two more bytes" never skips a candidate).
// So every position is converted independently.  Blocks start at arbitrary addresses: byte accesses.
__global__ __launch_bounds__(256) void bcj_simple_kernel(const semz_unit *units, uint32_t nunits, uint8_t *out,
                                                         const semz_result *results, int stage)
{
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const semz_unit un = units[u];
    const semz_result r = results[u];
    const uint32_t f = (un.filters >> (4 * stage)) & 15;
    if (r.status != SEMZ_OK || f < SEMZ_FILTER_POWERPC || f > SEMZ_FILTER_ARM64) return;
    uint8_t *b = out + r.out_off;
    const uint64_t n = r.out_len;
    const uint32_t now_pos = un.param2;
    const uint64_t step = f == SEMZ_FILTER_IA64 ? 16 : f == SEMZ_FILTER_ARMTHUMB ? 2 : 4;
    const uint64_t need = f == SEMZ_FILTER_IA64 ? 16 : 4;
    for (uint64_t i = step * threadIdx.x; i + need <= n; i += step * 256) {
        const uint32_t pc = now_pos + (uint32_t)i;
        if (f == SEMZ_FILTER_ARM) {
            if (b[i + 3] == 0xEB) {
                const uint32_t v = ((((uint32_t)b[i] | ((uint32_t)b[i + 1] << 8) | ((uint32_t)b[i + 2] << 16)) << 2) - (pc + 8)) >> 2;
                b[i] = (uint8_t)v; b[i + 1] = (uint8_t)(v >> 8); b[i + 2] = (uint8_t)(v >> 16);
            }
        } else if (f == SEMZ_FILTER_ARM64) {
            const uint32_t instr = (uint32_t)b[i] | ((uint32_t)b[i + 1] << 8) | ((uint32_t)b[i + 2] << 16) | ((uint32_t)b[i + 3] << 24);
            uint32_t v = instr;
            if ((instr >> 26) == 0x25) v = ((instr - (pc >> 2)) & 0x03FFFFFFu) | 0x94000000u;
            else if ((instr & 0x9F000000u) == 0x90000000u) {
                const uint32_t src = ((instr >> 29) & 3) | ((instr >> 3) & 0x001FFFFCu);
                if (((src + 0x00020000u) & 0x001C0000u) == 0) {
                    const uint32_t dest = src - (pc >> 12);
                    v = (instr & 0x9000001Fu) | ((dest & 3) << 29) | ((dest & 0x0003FFFCu) << 3) | ((0u - (dest & 0x00020000u)) & 0x00E00000u);
                }
            }
            if (v != instr) { b[i] = (uint8_t)v; b[i + 1] = (uint8_t)(v >> 8); b[i + 2] = (uint8_t)(v >> 16); b[i + 3] = (uint8_t)(v >> 24); }
        } else if (f == SEMZ_FILTER_ARMTHUMB) {
            if ((b[i + 1] & 0xF8) == 0xF0 && (b[i + 3] & 0xF8) == 0xF8) {
                const uint32_t src = ((((uint32_t)b[i + 1] & 7) << 19) | ((uint32_t)b[i] << 11) | (((uint32_t)b[i + 3] & 7) << 8) | (uint32_t)b[i + 2]) << 1;
                const uint32_t dest = (src - (pc + 4)) >> 1;
                b[i + 1] = (uint8_t)(0xF0 | ((dest >> 19) & 7));
                b[i] = (uint8_t)(dest >> 11);
                b[i + 3] = (uint8_t)(0xF8 | ((dest >> 8) & 7));
                b[i + 2] = (uint8_t)dest;
            }
        } else if (f == SEMZ_FILTER_POWERPC) {
            if ((b[i] >> 2) == 0x12 && (b[i + 3] & 3) == 1) {
                const uint32_t src = (((uint32_t)b[i] & 3) << 24) | ((uint32_t)b[i + 1] << 16) | ((uint32_t)b[i + 2] << 8) | ((uint32_t)b[i + 3] & ~3u);
                const uint32_t dest = src - pc;
                b[i] = (uint8_t)(0x48 | ((dest >> 24) & 3));
                b[i + 1] = (uint8_t)(dest >> 16);
                b[i + 2] = (uint8_t)(dest >> 8);
                b[i + 3] = (uint8_t)((b[i + 3] & 3) | (dest & 0xFF));
            }
        } else if (f == SEMZ_FILTER_SPARC) {
            if ((b[i] == 0x40 && (b[i + 1] & 0xC0) == 0x00) || (b[i] == 0x7F && (b[i + 1] & 0xC0) == 0xC0)) {
                const uint32_t src = ((uint32_t)b[i] << 24) | ((uint32_t)b[i + 1] << 16) | ((uint32_t)b[i + 2] << 8) | b[i + 3];
                const uint32_t dest = ((src << 2) - pc) >> 2;
                const uint32_t v = (((0u - ((dest >> 22) & 1)) << 22) & 0x3FFFFFFFu) | (dest & 0x3FFFFFu) | 0x40000000u;
                b[i] = (uint8_t)(v >> 24); b[i + 1] = (uint8_t)(v >> 16); b[i + 2] = (uint8_t)(v >> 8); b[i + 3] = (uint8_t)v;
            }
        } else {                                                   // IA-64: three 41-bit slots per 16-byte bundle
            const uint32_t tmpl = b[i] & 0x1F;
            // masks 16..31: 4 4 6 6 0 0 7 7 4 4 0 0 4 4 0 0 (three bits each), templates below 16 have none
            const uint64_t table = 0x0ull | (4ull << 0) | (4ull << 3) | (6ull << 6) | (6ull << 9) | (7ull << 18) | (7ull << 21) |
                                   (4ull << 24) | (4ull << 27) | (4ull << 36) | (4ull << 39);
            const uint32_t mask = tmpl < 16 ? 0u : (uint32_t)(table >> (3 * (tmpl - 16))) & 7;
            uint32_t bit_pos = 5;
            for (uint32_t slot = 0; slot < 3; slot++, bit_pos += 41) {
                if (!((mask >> slot) & 1)) continue;
                const uint64_t byte_pos = bit_pos >> 3;
                const uint32_t bit_res = bit_pos & 7;
                uint64_t instruction = 0;
                for (int k = 0; k < 6; k++) instruction |= (uint64_t)b[i + byte_pos + k] << (8 * k);
                const uint64_t inst_norm = instruction >> bit_res;
                if (((inst_norm >> 37) & 0xF) == 0x5 && ((inst_norm >> 9) & 0x7) == 0) {
                    const uint32_t src = (uint32_t)(((inst_norm >> 13) & 0xFFFFF) | (((inst_norm >> 36) & 1) << 20)) << 4;
                    const uint32_t dest = (src - pc) >> 4;
                    uint64_t v = (((uint64_t)1 << bit_res) - 1) & instruction;
                    v |= ((((uint64_t)(dest & 0x100000)) << (36 - 20)) | ((uint64_t)(dest & 0xFFFFF) << 13) |
                          (inst_norm & ~((uint64_t)0x8FFFFF << 13))) << bit_res;
                    for (int k = 0; k < 6; k++) b[i + byte_pos + k] = (uint8_t)(v >> (8 * k));
                }
            }
        }
    }
}

} // namespace

int semz_lzma2_handed_back(long *value)
{
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lz_handed_back), sizeof v) != hipSuccess) return -1;
    *value = (long)v;
    return 0;
}

int semz_run_lzma2(semz_ctx *c, const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                   uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails, bool wide)
{
    if (wide) {
        void *w = nullptr;
        int rc = semz_scratch(c, SEMZ_SLOT_LZ_WIDE, (size_t)nunits * WIDE_LIT_PROBS * sizeof(uint16_t), &w);
        if (rc) return rc;
        semz_span_begin(c, "lzma2_kernel");
        lzma2_kernel<true, false><<<nunits, 64, 0, c->stream>>>(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails, (uint16_t *)w, 0);
        semz_span_end(c);
    } else {
        // the accept-only fast kernel first; what it hands back (anything the reference would reject, unusual
        // properties, raw LZMA1) goes to the general kernel: eight units per CU, and the few units that need the
        // full literal table are redone by the third launch
        if (!c->lz_no_fast) {
            semz_span_begin(c, "lzma2_fast_kernel");
            semz_launch_lzma2_fast(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails, c->stream);
            semz_span_end(c);
        }
        semz_span_begin(c, "lzma2_kernel");
        lzma2_kernel<false, true><<<nunits, 64, 0, c->stream>>>(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails, nullptr, c->lz_no_fast ? 0 : 2);
        lzma2_kernel<false, false><<<nunits, 64, 0, c->stream>>>(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails, nullptr, 1);
        semz_span_end(c);
    }
    // filter chain stages (at most three non-last filters, xz.lisp:262); a kernel returns at once for
    // units whose chain has another filter, or none, at that stage
    semz_span_begin(c, "xz_filter_kernels");
    for (int stage = 0; stage < 3; stage++) {
        bcj_x86_kernel<<<nunits, 64, 0, c->stream>>>(d_units, nunits, d_out, d_results, stage);
        bcj_simple_kernel<<<nunits, 256, 0, c->stream>>>(d_units, nunits, d_out, d_results, stage);
        delta_kernel<<<nunits, 64, 0, c->stream>>>(d_units, nunits, d_out, d_results, stage);
    }
    semz_span_end(c);
    semz_span_begin(c, "checksum_kernel");
    semz_launch_checksum(d_out, d_units, nunits, d_results, d_tails, c->d_tabs,
                         (1u << SEMZ_TK_CRC32) | (1u << SEMZ_TK_CRC64) | (1u << SEMZ_TK_SHA256), c->stream);
    semz_span_end(c);
    return 0;
}
