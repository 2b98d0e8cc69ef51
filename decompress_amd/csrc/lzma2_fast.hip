// lzma2_fast.hip -- the LZMA2 / xz block decoder for well-formed units with ordinary properties.
//
// Same reference loops as lzma2.hip (lzma.lisp:49-198, 247-341, 566-775, lzma2.lisp:77-204,
// xz.lisp:426-457); this kernel only ever *accepts*.  Whatever it cannot vouch for -- any of the
// reference's failure conditions, literal / position properties outside lc + lp <= 3 and pb <= 2
// (xz writes lc 3, lp 0, pb 2), raw LZMA1 units, outputs of 2 GiB and more -- leaves the unit
// with LZ_RETRY_FULL, and the general kernel of lzma2.hip, launched right behind, decodes it
// again from the start and reports the reference's status and detail.  So the reject paths have
// one implementation, and this one is free to be organised for speed alone.
//
// What bounds an LZMA decoder on this chip is the number of instructions per decoded bit: a
// wave issues one instruction every eight clocks or so whatever its kind, and a
// batch of 1 MiB blocks has two waves per SIMD at most.  The general kernel spends about 47
// instructions per bit; this one spends about 15, and next to nothing between the bits:
//
//  * Probabilities live in vector registers, one model per lane, and a bit step takes its
//    probability with v_readlane and updates it in place (all lanes computed, one lane kept):
//    no LDS round trip, no address arithmetic.  The flag models, the length decoders' low / mid trees and the align
//    tree stay in registers for the whole unit; the 0x300-entry literal subtree of the current
//    context, the distance-slot tree, the distance trees of slots 4..13 and the length
//    decoders' high trees are fetched from LDS into registers when a symbol needs them (the
//    literal subtree speculatively, while the is-literal bit is being decoded) and written back
//    afterwards, whole registers at a time.
//  * Trees are walked in mirrored order: q' = 2 q + (bit == 0).  The comparison that decides the
//    bit leaves "code < bound" in SCC, so s_addc_u32 q, q, q is the whole tree step, and the three
//    selects of the range decoder hang on the same SCC.  Which lane holds which node is private
//    to this kernel (every model starts at 1024), the decoded value is ~q.
//  * The probability update is (31 p + K) >> 5 with K = 2048 or 31: equal to p + ((2048 - p) >> 5) and
//    p - (p >> 5) of lzma.lisp:104-112 (31 p + 2048 = 32 p + (2048 - p); 31 p + 31 = 32 p - (p - 31), and
//    ceil((p - 31) / 32) = floor(p / 32)), one multiply-add and one shift on the vector pipe.
//  * The check "code < range" after a normalisation (lzma.lisp:72-73) cannot fail once it held at
//    the start (code < range implies 256 code + byte < 256 range), so only the start and the one
//    case of the direct bits that can break the invariant (lzma.lisp:186-188) are tested.
//  * Input limits are tested once per symbol, not per byte: reads are clamped to the unit's last
//    dword, a symbol is applied only if the read position is still inside the chunk.
#include "semz_host.h"

namespace {

constexpr int32_t LZ_RETRY_FULL = -101;        // same value as in lzma2.hip
constexpr int F_RING = 4096, F_MASK = F_RING - 1;
constexpr int F_FLUSH = 1024;                  // 64 lanes x 16 bytes
constexpr int F_MAX_MATCH = 273;

typedef const __attribute__((address_space(4))) uint32_t *k_u32p;   // scalar-load view of read-only memory

struct __align__(16) FastLds {
    uint8_t ring[F_RING];
    uint16_t lit[0x300 * 8];                   // per context: [q] plain tree, [0x100 + (match bit << 8) + q] matched trees
    uint16_t slot[4 * 64];                     // [length state][q]
    uint16_t mid[10 * 64];                     // [slot - 4][q]   (reverse trees of one to five bits)
    uint16_t high[2 * 256];                    // [simple / rep][q]
    uint32_t stage[64];                        // where the bytes of a far match land (one dword per lane)
};
static_assert(sizeof(FastLds) <= 20480, "eight units per CU");

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// Placed right behind a lane-masked store: the two paths of the mask join here, in a block without phis (see inflate.hip).
__device__ __forceinline__ void join_lanes() { __builtin_amdgcn_wave_barrier(); }

// The range decoder's steps as assembler text, so that a run of bits is one block for the compiler: no copies of
// range / code / models between the bits, no spills.  With two waves on a SIMD it is the scalar unit that is
// saturated (one scalar instruction per ten clocks and wave, scripts/ubench/issue_rate.hip), the vector pipe has
// room: the probability update is done there, on all lanes, and kept for the one lane M0 points at (four vector
// instructions for four scalar ones and the v_writelane).  Operand names are the same in every block.
//   A_BIT(vt)   one modelled bit, the probability in lane M0 of vt; leaves SCC = (bit == 0) for the caller's
//               s_addc (A_QM0: q lives in M0; A_QREG: q is a register), then updates the model
//   A_NORM      lzma.lisp:64-73 with the byte taken from the input window (see Win)
#define A_BIT_HEAD(vt)                                                  \
    "v_readlane_b32 %[p], " vt ", m0\n\t"                               \
    "s_lshr_b32 %[b], %[range], 11\n\t"                                 \
    "s_lshl_b64 %[mq], 1, m0\n\t"                                       \
    "s_mul_i32 %[b], %[b], %[p]\n\t"                                    \
    "s_sub_u32 %[nr], %[range], %[b]\n\t"                               \
    "s_sub_u32 %[nc], %[code], %[b]\n\t" /* SCC = code < bound: 0 */    \
    "s_cselect_b32 %[range], %[b], %[nr]\n\t"                           \
    "s_cselect_b32 %[code], %[code], %[nc]\n\t"                         \
    "s_cselect_b32 %[k], 0x800, 31\n\t"
#define A_BIT_TAIL(vt)                                                  \
    "v_mad_u32_u24 %[t], " vt ", 31, %[k]\n\t"                          \
    "v_lshrrev_b32_e32 %[t], 5, %[t]\n\t"                               \
    "v_cndmask_b32_e64 " vt ", " vt ", %[t], %[mq]\n\t"
// (lzma.lisp:64-73.  One bit step in seven or so needs it, and a taken branch costs as much as three instructions
// (DESIGN.md 3.0): the common case falls through, the normalisation itself sits out of line in subsection 1 of the
// text section, each use with labels of its own -- %= numbers the asm statement, __COUNTER__ the use.)
#define A_NORM A_NORM_N(__COUNTER__, "")
#define A_NORM_N(n, first) A_NORM_S(n, first)
#define A_NORM_S(n, first)                                              \
    "s_cmp_lt_u32 %[range], 0x1000000\n\t"                              \
    "s_cbranch_scc1 .Lsemz_norm%=_" #n "\n"                             \
    ".Lsemz_back%=_" #n ":\n\t"                                         \
    ".subsection 1\n"                                                   \
    ".Lsemz_norm%=_" #n ":\n\t" first                                   \
    "s_lshr_b32 %[b], %[off], 2\n\t"                                    \
    "s_lshl_b32 %[range], %[range], 8\n\t"                              \
    "v_readlane_b32 %[k], %[win], %[b]\n\t"                             \
    "s_lshl_b32 %[b], %[off], 3\n\t"                                    \
    "s_lshl_b32 %[code], %[code], 8\n\t"                                \
    "s_lshr_b32 %[k], %[k], %[b]\n\t"                                   \
    "s_and_b32 %[k], %[k], 0xff\n\t"                                    \
    "s_add_u32 %[off], %[off], 1\n\t"                                   \
    "s_or_b32 %[code], %[code], %[k]\n\t"                               \
    "s_branch .Lsemz_back%=_" #n "\n\t"                                 \
    ".subsection 0\n\t"
// One direct bit (lzma.lisp:174-192): the range is halved, f' = 2 f + (bit == 0).  The reference fails when the code
// equals the halved range after a one bit.  That is the only way for "code < range" to break here, and once broken
// it stays broken through further halvings (c >= r gives c - (r >> 1) >= r >> 1); a normalisation could shift the
// evidence out of the 32 bits, so the test is made in front of every normalisation and once behind the last bit
// rather than at every bit.
#define A_DIRECT_BIT                                                    \
    "s_lshr_b32 %[range], %[range], 1\n\t"                              \
    "s_sub_u32 %[b], %[code], %[range]\n\t" /* SCC = code < range: 0 */ \
    "s_cselect_b32 %[code], %[code], %[b]\n\t"                          \
    "s_addc_u32 %[f], %[f], %[f]\n\t"                                   \
    A_NORM_N(__COUNTER__, "s_cmp_ge_u32 %[code], %[range]\n\ts_cmov_b32 %[bad], 1\n\t")
#define A_TREE(vt) A_BIT_HEAD(vt) "s_addc_u32 m0, m0, m0\n\t" A_BIT_TAIL(vt) A_NORM                                /* q in M0 */
#define A_TREEB(vt) "s_add_u32 m0, %[base], %[q]\n\t" A_BIT_HEAD(vt) "s_addc_u32 %[q], %[q], %[q]\n\t" A_BIT_TAIL(vt) A_NORM
// the last level of an eight-bit tree (q in M0, 128..255), whose 128 models fill two registers: q bit 6 says which
#define A_LEVEL7(va, vb)                                                \
    "v_readlane_b32 %[p], " va ", m0\n\t"                               \
    "v_readlane_b32 %[k], " vb ", m0\n\t"                               \
    "s_lshl_b64 %[mq], 1, m0\n\t"                                       \
    "s_bitcmp1_b32 m0, 6\n\t"                                           \
    "s_cselect_b32 %[p], %[k], %[p]\n\t"                                \
    "s_cselect_b64 %[mb], %[mq], 0\n\t"                                 \
    "s_andn2_b64 %[mq], %[mq], %[mb]\n\t"                               \
    "s_lshr_b32 %[b], %[range], 11\n\t"                                 \
    "s_mul_i32 %[b], %[b], %[p]\n\t"                                    \
    "s_sub_u32 %[nr], %[range], %[b]\n\t"                               \
    "s_sub_u32 %[nc], %[code], %[b]\n\t"                                \
    "s_cselect_b32 %[range], %[b], %[nr]\n\t"                           \
    "s_cselect_b32 %[code], %[code], %[nc]\n\t"                         \
    "s_cselect_b32 %[k], 0x800, 31\n\t"                                 \
    "s_addc_u32 m0, m0, m0\n\t"                                         \
    "v_mad_u32_u24 %[t], " va ", 31, %[k]\n\t"                          \
    "v_lshrrev_b32_e32 %[t], 5, %[t]\n\t"                               \
    "v_cndmask_b32_e64 " va ", " va ", %[t], %[mq]\n\t"                 \
    "v_mad_u32_u24 %[t], " vb ", 31, %[k]\n\t"                          \
    "v_lshrrev_b32_e32 %[t], 5, %[t]\n\t"                               \
    "v_cndmask_b32_e64 " vb ", " vb ", %[t], %[mb]\n\t" A_NORM
#define A_RC_OUT [range] "+s"(range), [code] "+s"(code), [off] "+s"(off), [p] "=&s"(p), [b] "=&s"(b), [nr] "=&s"(nr), \
                 [nc] "=&s"(nc), [k] "=&s"(k), [mq] "=&s"(mq), [t] "=&v"(t)
#define A_TEMPS uint32_t p, b, nr, nc, k, t; uint64_t mq

// ---- runs of LZMA symbols, decoded and applied by ONE block of assembler ----
// Between the bit runs above the compiler used to write the symbol's state machine: which flag comes next, which
// tree, the rep rotation, the state update.  With the models as read-write operands of many small blocks under C++
// control flow that was copies of model registers around every join, booleans kept as lane masks and spilled, and
// about 150 instructions of glue per symbol; and every return to the C++ loop cost another 170 or so (scalar state
// written to and read from lanes of a spill register).  The whole of decode-lzma (lzma.lisp:566-775) for the common
// symbols is now straight assembler that loops from symbol to symbol:
//  * the decision tree: is-literal; the literal's eight-level tree (after a literal: its context's subtree is read
//    from LDS at the top and written back at the end); is-simple, the length decoders with their high trees in LDS,
//    the slot tree, the reverse trees of slots 4..13, the direct bits and the align tree; the rep flags with the
//    rotation of rep0..rep3; the state table (lzma.lisp:615);
//  * a literal behind a match (lzma.lisp:497-518): the eight models on the match byte's path sit in lanes 0..7 of
//    one register (one LDS read), the walk follows them while the decoded bits agree with the match byte and goes
//    on in the plain tree from the level where they part;
//  * literals are applied (ring, position, context of the next symbol);
//  * matches shorter than 64 bytes whose distance is at least their length are applied without waiting for their
//    bytes: lane k asks for source byte k (ring: ds_read; older than the ring: the dword around it, HBM -> LDS
//    stage, see the kernel), and the bytes go into the ring when the next symbol needs them (A_RESOLVE).
// The block returns when the chunk's output is complete, the ring is due for a flush, the input window has to move
// (kind 0 behind a literal, 3 behind a match), or for a match it does not apply (kind 2, val = length; rep0 and
// state are updated, handle-match's checks lzma.lisp:600-612 are done).  `bad` != 0: one of those checks, the one
// failure of the direct bits (lzma.lisp:186-188) or an end marker; the caller hands the unit to the general kernel.
// Labels: 1 belongs to A_NORM; 5..99 are this block's.
#define A_FLAG(vt) A_BIT_HEAD(vt) "s_cselect_b32 %[f], 1, 0\n\t" A_BIT_TAIL(vt) A_NORM "s_cmp_eq_u32 %[f], 0\n\t"
#define A_TREE6(vt) A_TREE(vt) A_TREE(vt) A_TREE(vt) A_TREE(vt) A_TREE(vt) A_TREE(vt)
#define A_LDS_TREE8_IN(addr) "ds_read_u16 %[v0], " addr "\n\tds_read_u16 %[v1], " addr " offset:128\n\t" \
                             "ds_read_u16 %[v2], " addr " offset:256\n\tds_read_u16 %[v3], " addr " offset:384\n\t"
#define A_LDS_TREE8_OUT(addr) "ds_write_b16 " addr ", %[v0]\n\tds_write_b16 " addr ", %[v1] offset:128\n\t" \
                              "ds_write_b16 " addr ", %[v2] offset:256\n\tds_write_b16 " addr ", %[v3] offset:384\n\t"
#define A_TREE8 "s_mov_b32 m0, 1\n\t" A_TREE6("%[v0]") A_TREE("%[v1]") A_LEVEL7("%[v2]", "%[v3]")
// The bytes of the match that is still on its way (see the kernel) go into the ring; v0 = the bytes, one per lane.
#define A_RESOLVE                                                                                                   \
    "v_lshlrev_b32_e32 %[tv], 2, %[lane]\n\t"                                                                       \
    "s_waitcnt vmcnt(0)\n\t"                                                                                        \
    "ds_read_b32 %[v0], %[tv] offset:%[stageb]\n\t"                                                                 \
    "s_add_u32 %[x], %[ppend], %[shift]\n\t"                                                                        \
    "v_add_u32_e32 %[tv], %[x], %[lane]\n\t"                                                                        \
    "v_and_b32_e32 %[tv], 0xfff, %[tv]\n\t"                                                                         \
    "v_cmp_gt_u32_e32 vcc, 32, %[psh]\n\t"                                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                      \
    "v_lshrrev_b32_e32 %[v0], %[psh], %[v0]\n\t"                                                                    \
    "v_cndmask_b32_e32 %[v0], %[pl], %[v0], vcc\n\t"                                                                \
    "v_cmp_gt_u32_e32 vcc, %[plen], %[lane]\n\t"                                                                    \
    "s_and_saveexec_b64 %[mq], vcc\n\t"                                                                             \
    "ds_write_b8 %[tv], %[v0]\n\t"                                                                                  \
    "s_mov_b64 exec, %[mq]\n\t"
// decode-lzma-length (lzma.lisp:247-257): low tree and the two choices in w0, mid tree in w1, high tree in LDS at
// %[highb] + hoff; leaves the length code 0..271 in %[val].  la / lb / lc: three labels of this expansion.
#define A_LENGTH(w0, w1, hoff, la, lb, lc)                                                                          \
    "s_lshl_b32 %[base], %[ps], 3\n\t"                                                                              \
    "s_mov_b32 m0, 32\n\t" A_FLAG(w0)                                                                               \
    "s_cbranch_scc1 " la "f\n\t"                                                                                    \
    "s_mov_b32 %[q], 1\n\t" A_TREEB(w0) A_TREEB(w0) A_TREEB(w0)                                                     \
    "s_andn2_b32 %[val], 7, %[q]\n\t"                                                                               \
    "s_branch " lc "f\n"                                                                                            \
    la ":\n\t"                                                                                                      \
    "s_mov_b32 m0, 33\n\t" A_FLAG(w0)                                                                               \
    "s_cbranch_scc1 " lb "f\n\t"                                                                                    \
    "s_mov_b32 %[q], 1\n\t" A_TREEB(w1) A_TREEB(w1) A_TREEB(w1)                                                     \
    "s_andn2_b32 %[val], 7, %[q]\n\t"                                                                               \
    "s_add_u32 %[val], %[val], 8\n\t"                                                                               \
    "s_branch " lc "f\n"                                                                                            \
    lb ":\n\t"                                                                                                      \
    "v_add_u32_e32 %[tv], %[highb] + " hoff ", %[l2]\n\t"                                                            \
    A_LDS_TREE8_IN("%[tv]") "s_waitcnt lgkmcnt(0)\n\t" A_TREE8 A_LDS_TREE8_OUT("%[tv]")                             \
    "s_andn2_b32 %[val], 0xff, m0\n\t"                                                                              \
    "s_add_u32 %[val], %[val], 16\n"                                                                                \
    lc ":\n\t"

#ifdef SEMZ_STAMPS
// diagnostic build only (make STAMPS=1): shader-clock totals per phase summed over all units (scripts/lzma_stamps.py)
__device__ unsigned long long g_lz_stamps[16];
#define LZ_STAMP_DECL uint64_t t_last = __builtin_amdgcn_s_memtime(), t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint32_t n_acc[6] = {0, 0, 0, 0, 0, 0};
#define LZ_STAMP(i) do { uint64_t t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define LZ_COUNT(i) (n_acc[i]++)
#define WHY(i) do { if (lane == 0) atomicAdd(&g_lz_stamps[15], 1ull << (8 * (i))); } while (0)
#else
#define WHY(i) do { } while (0)
#define LZ_STAMP_DECL
#define LZ_STAMP(i) do { } while (0)
#define LZ_COUNT(i) do { } while (0)
#endif

struct Symbol { uint32_t kind, val, bad; };

// Position state of a run of symbols (decode_symbols): where the output stands and what ends the run.
struct Run {
    uint32_t p, total;             // output position, total-i-mod-16 counter (not reduced here)
    uint32_t prev, mbyte;          // last byte; behind a match: the byte that follows the match's source (both only if pend_len == 0)
    uint32_t endp, flushed;        // end of the chunk's output; bytes of the ring that are in HBM already
    uint32_t shift;                // ring index of output position 0
    uint32_t lc, lc8, lp_mask, pb_mask;   // lc, 8 - lc, masks of lp and pb
    uint32_t dict_size, dict_base; // lzma.lisp:600-612
    uint32_t ob_low;               // low two bits of the unit's output address
    uint64_t ob4;                  // that address rounded down to a dword
    // the far match whose bytes are on their way (see the kernel): length (0 = none), position, per lane the shift of
    // its byte inside the dword that was asked for (32 = the byte came from the ring) and that ring byte
    uint32_t pend_len, pend_p, pend_sh, pend_l;
};

__device__ __forceinline__ Symbol decode_symbols(uint32_t &range, uint32_t &code, uint32_t &off, uint32_t win, uint32_t &state,
                                                 uint32_t &rep0, uint32_t &rep1, uint32_t &rep2, uint32_t &rep3, Run &r,
                                                 uint32_t &vA, uint32_t &vB, uint32_t &vC, uint32_t &vS0, uint32_t &vS1,
                                                 uint32_t &vR0, uint32_t &vR1, uint32_t lane2, uint32_t lane)
{
    // (the kernel's LDS block starts at address 0, which the kernel checks once: the trees' addresses are immediates)
    constexpr uint32_t slotb = __builtin_offsetof(FastLds, slot), midb = __builtin_offsetof(FastLds, mid),
                       highb = __builtin_offsetof(FastLds, high), litb = __builtin_offsetof(FastLds, lit),
                       stageb = __builtin_offsetof(FastLds, stage);
    uint32_t ps, la;
    A_TEMPS;
    uint64_t mb;
    uint32_t kind, val, bad, f, x, y, base, q, v0, v1, v2, v3, tv, vm;
    asm volatile(
        "s_mov_b32 %[bad], 0\n"
        // ---- top of a symbol: pos_state, and (after a literal) the address of the context's literal subtree ----
        "5:\n\t"
        "s_and_b32 %[x], %[tot], %[lpm]\n\t"
        "s_lshr_b32 %[y], %[prev], %[lc8]\n\t"
        "s_lshl_b32 %[x], %[x], %[lc]\n\t"
        "s_or_b32 %[x], %[x], %[y]\n\t"
        "s_mulk_i32 %[x], 0x600\n\t"
        "s_and_b32 %[ps], %[tot], %[pbm]\n\t"
        "s_add_u32 %[x], %[x], %[litb]\n\t"
        "v_add_u32_e32 %[la], %[x], %[l2]\n\t"
        // ---- is-literal [state * 4 + pos_state]; after a literal the context's subtree is on its way meanwhile ----
        "s_lshl_b32 %[x], %[st], 2\n\t"
        "s_add_u32 m0, %[x], %[ps]\n\t"
        "s_cmp_gt_u32 %[st], 6\n\t"
        "s_cbranch_scc1 10f\n\t"
        A_LDS_TREE8_IN("%[la]")
        "10:\n\t"
        A_FLAG("%[vA]")
        "s_cbranch_scc1 20f\n\t"
        // ---- literal (lzma.lisp:661-688) ----
        "s_cmp_gt_u32 %[st], 6\n\t"
        "s_cbranch_scc1 70f\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        A_TREE8
        A_LDS_TREE8_OUT("%[la]")
        "s_andn2_b32 %[prev], 0xff, m0\n\t"              // the byte
        "s_sub_u32 %[x], %[st], 3\n\t"
        "s_cmp_lt_u32 %[st], 4\n\t"
        "s_cselect_b32 %[st], 0, %[x]\n"
        "15:\n\t"
        // the literal goes into the ring (every lane: same address, same value) and the next symbol follows at once
        // unless the chunk is complete, the ring is due for a flush or the input window has to move
        "s_add_u32 %[x], %[pp], %[shift]\n\t"
        "s_and_b32 %[x], %[x], 0xfff\n\t"
        "v_mov_b32_e32 %[tv], %[x]\n\t"
        "v_mov_b32_e32 %[v0], %[prev]\n\t"
        "s_add_u32 %[pp], %[pp], 1\n\t"
        "s_add_u32 %[tot], %[tot], 1\n\t"
        "ds_write_b8 %[tv], %[v0]\n\t"
        "s_mov_b32 %[kind], 0\n\t"
        "s_cmp_eq_u32 %[pp], %[endp]\n\t"
        "s_cbranch_scc1 99f\n\t"
        "s_sub_u32 %[x], %[pp], %[fl]\n\t"
        "s_cmp_ge_u32 %[x], 0x410\n\t"
        "s_cbranch_scc1 99f\n\t"
        "s_cmp_lt_u32 %[off], 192\n\t"
        "s_cbranch_scc1 5b\n\t"
        "s_branch 99f\n"
        // ---- a literal behind a match (lzma.lisp:497-518): the match byte's path is walked while the bits agree ----
        "70:\n\t"
        // the byte in front and the match byte: lanes length - 1 and length of a match whose bytes are still on their way
        "s_cmp_eq_u32 %[plen], 0\n\t"
        "s_cbranch_scc1 71f\n\t"
        A_RESOLVE
        "s_sub_u32 %[x], %[plen], 1\n\t"
        "v_readlane_b32 %[mbyte], %[v0], %[plen]\n\t"
        "v_readlane_b32 %[prev], %[v0], %[x]\n\t"
        "s_mov_b32 %[plen], 0\n\t"
        "s_and_b32 %[mbyte], %[mbyte], 0xff\n\t"
        "s_and_b32 %[prev], %[prev], 0xff\n"
        "71:\n\t"
        "s_and_b32 %[x], %[tot], %[lpm]\n\t"
        "s_lshr_b32 %[y], %[prev], %[lc8]\n\t"
        "s_lshl_b32 %[x], %[x], %[lc]\n\t"
        "s_or_b32 %[x], %[x], %[y]\n\t"
        "s_mulk_i32 %[x], 0x600\n\t"
        "s_add_u32 %[x], %[x], %[litb]\n\t"
        "v_add_u32_e32 %[la], %[x], %[l2]\n\t"
        A_LDS_TREE8_IN("%[la]")
        // lane l < 8: the model of level l on the match byte's path, [0x100 + (match bit << 8) + q] with the path's q,
        // which is the top of I = 0x100 | ~match byte (q counts zero bits as ones here)
        "s_andn2_b32 %[y], 0xff, %[mbyte]\n\t"
        "v_and_b32_e32 %[vm], 7, %[lane]\n\t"
        "s_or_b32 %[y], %[y], 0x100\n\t"
        "v_sub_u32_e32 %[vm], 8, %[vm]\n\t"
        "s_lshl_b32 %[f], %[mbyte], 1\n\t"
        "v_lshrrev_b32_e64 %[tv], %[vm], %[y]\n\t"
        "v_lshrrev_b32_e64 %[t], %[vm], %[f]\n\t"
        "s_add_u32 %[x], %[x], 0x200\n\t"
        "v_and_b32_e32 %[t], 1, %[t]\n\t"
        "v_lshl_add_u32 %[tv], %[t], 8, %[tv]\n\t"
        "v_lshl_add_u32 %[tv], %[tv], 1, %[x]\n\t"
        "ds_read_u16 %[vm], %[tv]\n\t"
        "s_mov_b32 %[q], 1\n\t"
        "s_mov_b32 %[x], 7\n\t"
        "s_mov_b32 m0, 0\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "72:\n\t"
        A_BIT_HEAD("%[vm]") "s_addc_u32 %[q], %[q], %[q]\n\t" A_BIT_TAIL("%[vm]") A_NORM
        "s_lshr_b32 %[f], %[y], %[x]\n\t"              // q of the path behind this level
        "s_add_u32 m0, m0, 1\n\t"
        "s_sub_u32 %[x], %[x], 1\n\t"
        "s_cmp_lg_u32 %[q], %[f]\n\t"
        "s_cbranch_scc1 73f\n\t"
        "s_cmp_lt_u32 m0, 8\n\t"
        "s_cbranch_scc1 72b\n"
        "73:\n\t"
        "s_mov_b64 exec, 0xff\n\t"
        "ds_write_b16 %[tv], %[vm]\n\t"
        "s_mov_b64 exec, -1\n\t"
        // the plain tree from the level the walk has reached (m0) on: levels 0..5 in v0, 6 in v1, 7 in v2 / v3
        "s_mov_b32 %[x], m0\n\t"
        "s_mov_b32 m0, %[q]\n\t"
        "s_cmp_gt_u32 %[x], 7\n\t"
        "s_cbranch_scc1 77f\n"
        "74:\n\t"
        "s_cmp_gt_u32 %[x], 5\n\t"
        "s_cbranch_scc1 75f\n\t"
        A_TREE("%[v0]")
        "s_add_u32 %[x], %[x], 1\n\t"
        "s_branch 74b\n"
        "75:\n\t"
        "s_cmp_gt_u32 %[x], 6\n\t"
        "s_cbranch_scc1 76f\n\t"
        A_TREE("%[v1]")
        "76:\n\t"
        A_LEVEL7("%[v2]", "%[v3]")
        A_LDS_TREE8_OUT("%[la]")
        "77:\n\t"
        "s_andn2_b32 %[prev], 0xff, m0\n\t"              // the byte
        "s_sub_u32 %[x], %[st], 3\n\t"
        "s_sub_u32 %[y], %[st], 6\n\t"
        "s_cmp_lt_u32 %[st], 10\n\t"
        "s_cselect_b32 %[st], %[x], %[y]\n\t"            // lzma.lisp:615 (state >= 7 here)
        "s_branch 15b\n"
        // ---- match: is-simple [state] ----
        "20:\n\t"
        "s_mov_b32 m0, %[st]\n\t"
        A_FLAG("%[vC]")
        "s_cbranch_scc1 40f\n\t"
        // simple match: length, then distance (lzma.lisp:323-341)
        A_LENGTH("%[vS0]", "%[vS1]", "0", "21", "22", "23")
        "s_cmp_lt_u32 %[st], 7\n\t"
        "s_cselect_b32 %[st], 7, 10\n\t"
        "s_mov_b32 %[r3], %[r2]\n\t"
        "s_mov_b32 %[r2], %[r1]\n\t"
        "s_mov_b32 %[r1], %[r0]\n\t"
        "s_min_u32 %[x], %[val], 3\n\t"
        "s_lshl_b32 %[x], %[x], 7\n\t"
        "s_add_u32 %[x], %[x], %[slotb]\n\t"
        "v_add_u32_e32 %[tv], %[x], %[l2]\n\t"
        "ds_read_u16 %[v0], %[tv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 m0, 1\n\t"
        A_TREE6("%[v0]")
        "ds_write_b16 %[tv], %[v0]\n\t"
        "s_andn2_b32 %[x], 63, m0\n\t"                   // the slot
        "s_mov_b32 %[r0], %[x]\n\t"
        "s_cmp_lt_u32 %[x], 4\n\t"
        "s_cbranch_scc1 30f\n\t"
        "s_lshr_b32 %[y], %[x], 1\n\t"
        "s_sub_u32 %[y], %[y], 1\n\t"                    // nb
        "s_and_b32 %[r0], %[x], 1\n\t"
        "s_or_b32 %[r0], %[r0], 2\n\t"
        "s_lshl_b32 %[r0], %[r0], %[y]\n\t"              // min_dist
        "s_cmp_lt_u32 %[x], 14\n\t"
        "s_cbranch_scc0 27f\n\t"
        // slots 4..13: a reverse tree of nb = 1..5 bits, least significant first
        "s_sub_u32 %[x], %[x], 4\n\t"
        "s_lshl_b32 %[x], %[x], 7\n\t"
        "s_add_u32 %[x], %[x], %[midb]\n\t"
        "v_add_u32_e32 %[tv], %[x], %[l2]\n\t"
        "ds_read_u16 %[v1], %[tv]\n\t"
        "s_mov_b32 %[x], %[y]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 m0, 1\n"
        "25:\n\t"
        A_TREE("%[v1]")
        "s_sub_u32 %[x], %[x], 1\n\t"
        "s_cmp_lg_u32 %[x], 0\n\t"
        "s_cbranch_scc1 25b\n\t"
        "ds_write_b16 %[tv], %[v1]\n\t"
        "s_not_b32 %[x], m0\n\t"
        "s_bfm_b32 %[f], %[y], 0\n\t"
        "s_and_b32 %[x], %[x], %[f]\n\t"
        "s_brev_b32 %[x], %[x]\n\t"
        "s_sub_u32 %[f], 32, %[y]\n\t"
        "s_lshr_b32 %[x], %[x], %[f]\n\t"
        "s_add_u32 %[r0], %[r0], %[x]\n\t"
        "s_branch 30f\n"
        // slots 14..63: nb - 4 direct bits (lzma.lisp:174-192), then the four align bits
        "27:\n\t"
        "s_sub_u32 %[x], %[y], 4\n\t"                    // 2..26 bits: one if the count is odd, then pairs
        "s_mov_b32 %[f], 0\n\t"
        "s_lshr_b32 %[base], %[x], 1\n\t"
        "s_sub_u32 %[base], %[base], 1\n\t"
        "s_lshl_b32 %[base], 1, %[base]\n\t"             // one bit per pair, shifted out below
        "s_bitcmp1_b32 %[x], 0\n\t"
        "s_cbranch_scc0 28f\n\t"
        A_DIRECT_BIT
        "28:\n\t"
        A_DIRECT_BIT
        A_DIRECT_BIT
        "s_lshr_b32 %[base], %[base], 1\n\t"             // SCC = pairs are left
        "s_cbranch_scc1 28b\n\t"
        "s_cmp_ge_u32 %[code], %[range]\n\t"
        "s_cmov_b32 %[bad], 1\n\t"
        "s_sub_u32 %[x], %[y], 4\n\t"
        "s_bfm_b32 %[x], %[x], 0\n\t"
        "s_andn2_b32 %[f], %[x], %[f]\n\t"
        "s_lshl_b32 %[f], %[f], 4\n\t"
        "s_add_u32 %[r0], %[r0], %[f]\n\t"
        "s_mov_b32 %[base], 48\n\t"
        "s_mov_b32 %[q], 1\n\t"
        A_TREEB("%[vC]") A_TREEB("%[vC]") A_TREEB("%[vC]") A_TREEB("%[vC]")
        "s_andn2_b32 %[x], 15, %[q]\n\t"
        "s_brev_b32 %[x], %[x]\n\t"
        "s_lshr_b32 %[x], %[x], 28\n\t"
        "s_add_u32 %[r0], %[r0], %[x]\n"
        "30:\n\t"
        "s_cmp_eq_u32 %[r0], -1\n\t"                     // an end marker is illegal in LZMA2 (lzma.lisp:727-732)
        "s_cmov_b32 %[bad], 2\n\t"
        "s_add_u32 %[val], %[val], 2\n\t"
        "s_branch 60f\n"
        // ---- rep matches: uses-rep0 [12 + state] ----
        "40:\n\t"
        "s_add_u32 m0, %[st], 12\n\t"
        A_FLAG("%[vC]")
        "s_cbranch_scc1 44f\n\t"
        "s_lshl_b32 %[x], %[st], 2\n\t"
        "s_add_u32 m0, %[x], %[ps]\n\t"
        A_FLAG("%[vB]")                                  // is-short-rep [state * 4 + pos_state]
        "s_cbranch_scc1 47f\n\t"
        "s_cmp_lt_u32 %[st], 7\n\t"
        "s_cselect_b32 %[st], 9, 11\n\t"
        "s_mov_b32 %[val], 1\n\t"
        "s_branch 60f\n"
        "44:\n\t"
        "s_add_u32 m0, %[st], 24\n\t"
        A_FLAG("%[vC]")                                  // is-rep1 [24 + state]
        "s_cbranch_scc1 45f\n\t"
        "s_mov_b32 %[x], %[r1]\n\t"
        "s_mov_b32 %[r1], %[r0]\n\t"
        "s_mov_b32 %[r0], %[x]\n\t"
        "s_branch 47f\n"
        "45:\n\t"
        "s_add_u32 m0, %[st], 36\n\t"
        A_FLAG("%[vC]")                                  // is-rep2 [36 + state]
        "s_cbranch_scc1 46f\n\t"
        "s_mov_b32 %[x], %[r2]\n\t"
        "s_mov_b32 %[r2], %[r1]\n\t"
        "s_mov_b32 %[r1], %[r0]\n\t"
        "s_mov_b32 %[r0], %[x]\n\t"
        "s_branch 47f\n"
        "46:\n\t"
        "s_mov_b32 %[x], %[r3]\n\t"
        "s_mov_b32 %[r3], %[r2]\n\t"
        "s_mov_b32 %[r2], %[r1]\n\t"
        "s_mov_b32 %[r1], %[r0]\n\t"
        "s_mov_b32 %[r0], %[x]\n"
        "47:\n\t"
        "s_cmp_lt_u32 %[st], 7\n\t"
        "s_cselect_b32 %[st], 8, 11\n\t"
        A_LENGTH("%[vR0]", "%[vR1]", "512", "41", "42", "43")
        "s_add_u32 %[val], %[val], 2\n"
        // ---- handle-match (lzma.lisp:600-612): length in val, distance - 1 in rep0 ----
        "60:\n\t"
        "s_cmp_ge_u32 %[r0], %[dsz]\n\t"
        "s_cmov_b32 %[bad], 3\n\t"
        "s_sub_u32 %[x], %[pp], %[dbase]\n\t"
        "s_cmp_ge_u32 %[r0], %[x]\n\t"
        "s_cmov_b32 %[bad], 4\n\t"
        "s_sub_u32 %[x], %[endp], %[pp]\n\t"
        "s_cmp_gt_u32 %[val], %[x]\n\t"
        "s_cmov_b32 %[bad], 5\n\t"
        "s_mov_b32 %[kind], 2\n\t"
        "s_cmp_lg_u32 %[bad], 0\n\t"
        "s_cbranch_scc1 99f\n\t"
        // The copy is done right here when it is the common kind: fewer than 64 bytes whose source reaches behind
        // the ring (see the kernel: asked for from HBM, not waited for).  A ring that is due for a flush, a source
        // inside the ring (source + 3823 > p + length) and longer matches go back to the C++ code as kind 2.
        "s_sub_u32 %[x], %[pp], %[fl]\n\t"
        "s_cmp_ge_u32 %[x], 0x410\n\t"
        "s_cbranch_scc1 99f\n\t"
        "s_cmp_lt_u32 %[r0], %[val]\n\t"
        "s_cbranch_scc1 99f\n\t"
        "s_cmp_gt_u32 %[val], 63\n\t"
        "s_cbranch_scc1 99f\n\t"
        // (a literal subtree asked for at the top of the symbol may still be on its way into v0..v3)
        "s_waitcnt lgkmcnt(0)\n\t"
        // the bytes of the match before this one, if they are still on their way, go into the ring first
        "s_cmp_eq_u32 %[plen], 0\n\t"
        "s_cbranch_scc1 61f\n\t"
        A_RESOLVE
        "s_mov_b32 %[plen], 0\n"
        "61:\n\t"
        // lane k <= length takes source byte k (lane `length`: the byte behind the source, what a literal right
        // behind the match is matched against); bytes the ring still holds are taken from there
        "s_sub_u32 %[x], %[pp], %[r0]\n\t"
        "s_sub_u32 %[x], %[x], 1\n\t"
        "v_cmp_ge_u32_e32 vcc, %[val], %[lane]\n\t"
        "v_cndmask_b32_e32 %[tv], 0, %[lane], vcc\n\t"
        "v_add_u32_e32 %[tv], %[x], %[tv]\n\t"           // source position of this lane
        "v_add_u32_e32 %[v1], %[obl], %[tv]\n\t"
        "v_add_u32_e32 %[v3], 0xeef, %[tv]\n\t"
        "v_and_b32_e32 %[v2], 3, %[v1]\n\t"
        "v_cmp_ge_u32_e32 vcc, %[pp], %[v3]\n\t"         // in HBM only: position + 3823 <= p
        "v_lshlrev_b32_e32 %[v2], 3, %[v2]\n\t"
        "v_and_b32_e32 %[v1], -4, %[v1]\n\t"
        "v_cndmask_b32_e32 %[psh], 32, %[v2], vcc\n\t"
        "v_cndmask_b32_e32 %[v1], 0, %[v1], vcc\n\t"
        "v_add_u32_e32 %[tv], %[shift], %[tv]\n\t"
        "v_and_b32_e32 %[tv], 0xfff, %[tv]\n\t"
        "ds_read_u8 %[pl], %[tv]\n\t"
        "s_add_u32 %[x], %[r0], %[val]\n\t"
        "s_cmp_lt_u32 %[x], 3822\n\t"                  // every byte is in the ring
        "s_cbranch_scc1 62f\n\t"
        "s_mov_b32 m0, %[stageb]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %[v1], %[ob4] sc1\n"
        "62:\n\t"
        "s_mov_b32 %[plen], %[val]\n\t"
        "s_mov_b32 %[ppend], %[pp]\n\t"
        "s_add_u32 %[pp], %[pp], %[val]\n\t"
        "s_add_u32 %[tot], %[tot], %[val]\n\t"
        "s_mov_b32 %[kind], 3\n\t"
        "s_cmp_eq_u32 %[pp], %[endp]\n\t"
        "s_cbranch_scc1 99f\n\t"
        "s_cmp_lt_u32 %[off], 192\n\t"
        "s_cbranch_scc1 5b\n"
        "99:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [vA] "+v"(vA), [vB] "+v"(vB), [vC] "+v"(vC), [vS0] "+v"(vS0), [vS1] "+v"(vS1), [vR0] "+v"(vR0), [vR1] "+v"(vR1),
          [st] "+s"(state), [r0] "+s"(rep0), [r1] "+s"(rep1), [r2] "+s"(rep2), [r3] "+s"(rep3), A_RC_OUT, [mb] "=&s"(mb),
          [kind] "=&s"(kind), [val] "=&s"(val), [bad] "=&s"(bad), [f] "=&s"(f), [x] "=&s"(x), [y] "=&s"(y), [base] "=&s"(base),
          [q] "=&s"(q), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [tv] "=&v"(tv)
          , [pp] "+s"(r.p), [tot] "+s"(r.total), [prev] "+s"(r.prev), [ps] "=&s"(ps), [la] "=&v"(la), [vm] "=&v"(vm), [mbyte] "+s"(r.mbyte),
          [plen] "+s"(r.pend_len), [ppend] "+s"(r.pend_p), [psh] "+v"(r.pend_sh), [pl] "+v"(r.pend_l)
        : [win] "v"(win), [l2] "v"(lane2), [slotb] "i"(slotb), [midb] "i"(midb), [highb] "i"(highb), [litb] "i"(litb),
          [endp] "s"(r.endp), [fl] "s"(r.flushed), [shift] "s"(r.shift), [lc] "s"(r.lc), [lc8] "s"(r.lc8), [lpm] "s"(r.lp_mask),
          [pbm] "s"(r.pb_mask), [dsz] "s"(r.dict_size), [dbase] "s"(r.dict_base), [obl] "s"(r.ob_low), [ob4] "s"(r.ob4),
          [lane] "v"(lane), [stageb] "i"(stageb)
        : "scc", "vcc", "m0", "memory");
    return Symbol{kind, val, bad};
}

// Input: 256 bytes of the unit's compressed data sit in one register, a dword per lane; `off` counts bytes from
// the window's start.  A normalisation takes its byte with v_readlane and two shifts, inside one asm block with
// its own branch: the compiler sees straight-line code (no phi webs around every bit step, no spills).  The
// window is moved forward between symbols (a symbol takes fewer than 64 bytes).
struct Win {
    const uint32_t *base;  // dword view of the input from the aligned address below the unit's first byte
    uint32_t wbase;        // offset (bytes from base, a multiple of four) of the window's first byte
    uint32_t off;          // next byte, relative to wbase
    uint32_t last;         // index of the last dword that holds a byte of the unit's input span
    uint32_t v;            // lane l: dword wbase / 4 + l (clamped to `last`)
    __device__ __forceinline__ uint32_t pos() const { return wbase + off; }
    __device__ __forceinline__ void seek(uint32_t o)
    {
        wbase = o & ~3u;
        off = o & 3;
        const uint32_t d = (wbase >> 2) + (threadIdx.x & 63);
        v = base[d < last ? d : last];
        asm volatile("" : "+v"(v));     // waited for here, not at the next bit step: a far match's bytes may be in flight there
    }
    __device__ __forceinline__ void slide()
    {
        if (off >= 192) seek(pos());
    }
    __device__ __forceinline__ uint32_t byte()             // headers only: the symbol loop uses normq()
    {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(off >> 2));
        const uint32_t b = (w >> (8 * (off & 3))) & 0xFF;
        off++;
        return b;
    }
};

__global__ __launch_bounds__(64, 2) void lzma2_fast_kernel(const uint8_t *in, uint64_t in_len, const semz_unit *units,
                                                           uint32_t nunits, uint8_t *out, uint64_t out_total,
                                                           semz_result *results, semz_tail *tails)
{
    __shared__ FastLds L;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
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
    // everything this kernel does not take goes to the general kernel
    auto retry = [&]() {
        if (lane == 0) {
            semz_result r{};
            r.status = LZ_RETRY_FULL;
            r.out_off = un.out_off;
            results[u] = r;
            tails[u] = semz_tail{};
        }
    };
    const bool is_xz = un.kind == SEMZ_UNIT_XZ_BLOCK;
    if ((un.kind != SEMZ_UNIT_XZ_BLOCK && un.kind != SEMZ_UNIT_LZMA2) || un.out_cap >= 0x7FF00000ull ||
        un.in_limit - un.in_off >= 0x7FF00000ull) {
        retry();
        return;
    }
    uint8_t *ob = out + un.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)ob & 15);
    const uint32_t out_cap = (uint32_t)un.out_cap;
    const uint32_t dict_size = un.param0;
    uint8_t *safe_byte = out;

    // input window: offsets are relative to the aligned address below the unit's first byte
    const uint8_t *ustart = in + un.in_off;
    const uint32_t amis = (uint32_t)((uintptr_t)ustart & 3);
    Win s;
    s.base = reinterpret_cast<const uint32_t *>(ustart - amis);
    const uint32_t phys_end = (uint32_t)(un.in_limit - un.in_off) + amis;           // first offset past the input
    uint32_t block_end = 0xFFFFFFFFu;                                                // declared xz compressed size
    if (is_xz && (un.flags & SEMZ_UF_HAVE_IN_SIZE) && un.size_in < 0x7FF00000ull) block_end = (uint32_t)un.size_in + amis;
    if (phys_end == amis) { retry(); return; }                                       // empty input: eof
    s.last = (phys_end - 1) >> 2;
    s.seek(amis);
    const uint32_t lim = phys_end < block_end ? phys_end : block_end;                // no read may pass this

    // ---- models that stay in registers (one model per lane) ----
    // vA   is-literal   [state * 4 + pos_state]
    // vB   is-short-rep [state * 4 + pos_state]
    // vC   is-simple [state], uses-rep0 [12 + state], is-rep1 [24 + state], is-rep2 [36 + state], align [48 + q]
    // vS0  simple-match length: low [pos_state * 8 + q], choice 32, choice-2 33;  vS1  mid [pos_state * 8 + q]
    // vR0, vR1 the same for rep-match lengths
    uint32_t vA = 1024, vB = 1024, vC = 1024, vS0 = 1024, vS1 = 1024, vR0 = 1024, vR1 = 1024;

    uint32_t range = 0, code = 0;
    uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0, state = 0, total = 0;
    uint32_t lc = 0, lp_mask = 0, pb_mask = 0;
    bool have_state = false;
    enum { NEED_DICT = 0, NEED_PROPS = 1, NEED_NONE = 2 };
    int need = NEED_DICT;
    uint32_t p = 0, dict_base = 0, flushed = 0;
    bool ok = true;                 // false: hand the unit to the general kernel

    LZ_STAMP_DECL
    auto flush = [&](bool all) {
        semz_take_turn(16);                      // (semz_device.h: the two units of a SIMD end together; 0.65 ms a turn)
        const uint32_t head = (16 - shift) & 15;
        if (flushed == 0 && head && (p >= head || all)) {
            const uint32_t n = p < head ? p : head;
            if (lane < n) ob[lane] = L.ring[(lane + shift) & F_MASK];
            join_lanes();
            flushed = n;
        }
        flushed = uni(flushed);
        if (flushed || !head) {
            while (p - flushed >= (uint32_t)F_FLUSH) {
                const uint32_t a = flushed + 16u * lane;
                const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(a + shift) & F_MASK]);
                *reinterpret_cast<uint4 *>(ob + a) = v;
                flushed += F_FLUSH;
            }
            if (all) {
                const uint32_t pieces = (p - flushed) >> 4;
                if (lane < pieces) {
                    const uint32_t a = flushed + 16u * lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(a + shift) & F_MASK]);
                    *reinterpret_cast<uint4 *>(ob + a) = v;
                }
                join_lanes();
                flushed += 16u * pieces;
                const uint32_t rest = p - flushed;
                if (lane < rest) ob[flushed + lane] = L.ring[(flushed + lane + shift) & F_MASK];
                join_lanes();
                flushed = p;
            }
        }
        flushed = uni(flushed);
    };
    // byte of history at position pos < p (every lane gets it)
    auto history_byte_v = [&](uint32_t pos) -> uint32_t {
        if (pos + (F_RING - F_MAX_MATCH) > p) return (uint32_t)L.ring[(pos + shift) & F_MASK];
        return (uint32_t)__hip_atomic_load(ob + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
#define PIN_ALL()                                                                                                       \
    do {                                                                                                                \
        p = uni(p); total = uni(total); prev = uni(prev); flushed = uni(flushed);                                       \
        dict_base = uni(dict_base); s.wbase = uni(s.wbase);                                                             \
    } while (0)

    // A match whose source lies behind the ring is not waited for: its bytes are asked for (HBM, one round trip) and
    // put into the ring when something needs them -- the next literal (context and match byte), the next match's
    // copy, a flush, the end of the chunk.  The symbol after a match is a match three times out of four, and its
    // bits do not depend on the bytes.  Lane k < pend_len holds byte k, lane pend_len the byte behind the source
    // (what a literal right behind the match is matched against).
    uint32_t pend_len = 0, pend_p = 0, pend_sh = 0, pend_l = 0;
    const uint32_t lane2 = 2 * lane;
    const uint32_t ob_low = (uint32_t)((uintptr_t)ob & 3);
    const uint64_t ob4 = (uint64_t)(uintptr_t)ob - ob_low;
    if (uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)L.ring) != 0) { retry(); return; }   // (see decode_symbols)
    auto resolve = [&]() -> uint32_t {                    // returns the bytes, one per lane
        LZ_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the load below is not one the compiler knows of)
        LZ_STAMP(1);
        const uint32_t w = L.stage[lane];
        const uint32_t bv = pend_sh < 32 ? (w >> pend_sh) & 0xFF : pend_l;
        if (lane < pend_len) L.ring[(pend_p + lane + shift) & F_MASK] = (uint8_t)bv;
        join_lanes();
        pend_len = 0;
        return bv;
    };
    // ---- LZMA2 control loop (lzma2.lisp:77-204) ----
    for (;;) {
        // (pins: what the optimiser merges behind the lane-masked stores of flush() and the copies counts as divergent
        // in its analysis, and every model index has to be a scalar)
        lc = uni(lc); lp_mask = uni(lp_mask); pb_mask = uni(pb_mask); p = uni(p); total = uni(total); state = uni(state);
        dict_base = uni(dict_base); flushed = uni(flushed); s.off = uni(s.off); s.wbase = uni(s.wbase);
        s.slide();
        rep0 = uni(rep0); rep1 = uni(rep1); rep2 = uni(rep2); rep3 = uni(rep3);
        if (s.pos() >= lim) { ok = false; break; }
        const uint32_t control = s.byte();
        if (control == 0) break;                                     // end of the LZMA2 stream
        if (control < 0x80) {
            if (control > 2) { ok = false; break; }
            if (s.pos() + 2 > lim) { ok = false; break; }
            const uint32_t b0 = s.byte(), b1 = s.byte();
            const uint32_t usize = ((b0 << 8) | b1) + 1;
            if (control == 2) {
                if (need == NEED_DICT) { ok = false; break; }
                need = NEED_NONE;
            } else {
                need = NEED_PROPS;
                dict_base = p;
            }
            if (usize > out_cap - p || usize > lim - s.pos()) { ok = false; break; }
            const uint8_t *src = ustart + (s.pos() - amis);
            lds_sync();
            for (uint32_t o = 0; o < usize; o += F_FLUSH) {
                const uint32_t n = usize - o < (uint32_t)F_FLUSH ? usize - o : (uint32_t)F_FLUSH;
                for (uint32_t k0 = 0; k0 < n; k0 += 64) {            // (uniform trip count: the decoder state stays scalar)
                    const uint32_t k = k0 + lane;
                    if (k < n) L.ring[(p + k + shift) & F_MASK] = src[o + k];
                    join_lanes();
                }
                p += n;
                lds_sync();
                flush(false);
            }
            s.seek(s.pos() + usize);
            if (have_state) total = (total + usize) & 15;            // lzma2.lisp:184-188
            continue;
        }
        if (s.pos() + 4 > lim) { ok = false; break; }
        const uint32_t b0 = s.byte(), b1 = s.byte(), b2 = s.byte(), b3 = s.byte();
        const uint32_t usize = (((control & 31) << 16) | (b0 << 8) | b1) + 1;
        const uint32_t csize = ((b2 << 8) | b3) + 1;
        if ((need == NEED_DICT && control < 0xE0) || (need == NEED_PROPS && control < 0xC0)) { ok = false; break; }
        need = NEED_NONE;
        if (control >= 0xC0) {
            if (s.pos() >= lim) { ok = false; break; }
            const uint32_t octet = s.byte();                         // parse-lzma2-props, lzma2.lisp:31-38
            const uint32_t nlc = octet % 9, nlp = (octet / 9) % 5, npb = octet / 45;
            if (npb > 2 || nlc + nlp > 3) { ok = false; break; }     // (including the illegal ones)
            lc = nlc; lp_mask = (1u << nlp) - 1; pb_mask = (1u << npb) - 1;
        } else if (!have_state) { ok = false; break; }
        if (control >= 0xE0) dict_base = p;
        if (usize > out_cap - p) { ok = false; break; }
        if (csize > lim - s.pos() || csize < 5) { ok = false; break; }
        const uint32_t chunk_end = s.pos() + csize;
        const uint32_t endp = p + usize;
        // the range decoder (lzma.lisp:55-62)
        if (s.byte() != 0) { ok = false; break; }
        range = 0xFFFFFFFFu;
        code = 0;
        for (int i = 0; i < 4; i++) code = (code << 8) | s.byte();
        if (code == 0xFFFFFFFFu) { ok = false; break; }
        if (control >= 0xA0 || !have_state) {                        // fresh lzma-state-vars (lzma2.lisp:144-145)
            vA = vB = vC = vS0 = vS1 = vR0 = vR1 = 1024;
            {
                uint32_t *w32 = reinterpret_cast<uint32_t *>(L.lit);
                constexpr uint32_t NW = (sizeof(L.lit) + sizeof(L.slot) + sizeof(L.mid) + sizeof(L.high)) / 4;
                static_assert(NW % 64 == 0, "whole passes of the wave");
                for (uint32_t i0 = 0; i0 < NW; i0 += 64) w32[i0 + lane] = 0x04000400u;
            }
            rep0 = rep1 = rep2 = rep3 = 0; state = 0; total = 0;
            lds_sync();
        }
        have_state = true;
        // the byte in front of the chunk, for the first literal's context, and (behind a match) the match byte
        uint32_t prev = p == dict_base ? 0u : uni((uint32_t)L.ring[(p - 1 + shift) & F_MASK]);
        uint32_t mbyte = 0;
        if (state >= 7) {
            if (p - dict_base < rep0 + 1) { ok = false; break; }
            mbyte = uni(history_byte_v(p - 1 - rep0));
        }

        // ---- decode-lzma (lzma.lisp:566-775) ----
        while (p != endp) {
            PIN_ALL();
            s.slide();
            Run run{p, total, prev, mbyte, endp, flushed, shift, lc, 8 - lc, lp_mask, pb_mask, dict_size, dict_base, ob_low, ob4,
                    pend_len, pend_p, pend_sh, pend_l};
            LZ_STAMP(7);
            const Symbol sym = decode_symbols(range, code, s.off, s.v, state, rep0, rep1, rep2, rep3, run, vA, vB, vC, vS0, vS1, vR0,
                                              vR1, lane2, lane);
            p = run.p; total = run.total; prev = run.prev; mbyte = run.mbyte;
            pend_len = run.pend_len; pend_p = run.pend_p; pend_sh = run.pend_sh; pend_l = run.pend_l;
            LZ_STAMP(0);
            LZ_COUNT(sym.kind);
            if (sym.bad) { ok = false; WHY(sym.bad); break; }
            if (s.pos() > chunk_end) { ok = false; WHY(2); break; }
            // whatever ended the run, the ring may be due for a flush; the bytes of a match that are still on their
            // way go in first (a literal may follow: it needs the match's last byte and its match byte)
            if (p - flushed >= (uint32_t)F_FLUSH + 16) {
                if (pend_len) {
                    const uint32_t n = uni(pend_len);
                    const uint32_t bv = resolve();
                    prev = (uint32_t)__builtin_amdgcn_readlane((int)bv, (int)(n - 1)) & 0xFF;
                    mbyte = (uint32_t)__builtin_amdgcn_readlane((int)bv, (int)n) & 0xFF;
                }
                lds_sync();
                LZ_STAMP(2);
                flush(false);
                LZ_STAMP(3);
            }
            if (sym.kind != 2) {                 // a run of literals and short matches, applied already
                LZ_STAMP(2);
                continue;
            }
            // ---- a match the assembler block does not apply: 64 bytes or more, or closer than it is long ----
            const uint32_t len = sym.val;
            // (handle-match, lzma.lisp:600-612, was checked in decode_symbols)
            if (pend_len) resolve();
            lds_sync();
            LZ_STAMP(4);
            const uint32_t dist = rep0 + 1;
            const uint32_t src = p - dist;
            if (src + (F_RING - F_MAX_MATCH) > p + len) {
                // copy-match (common.lisp:324-333), all lanes: all of the source is in the ring
                if (dist >= len) {
                    for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        const uint32_t bv = L.ring[(src + k + shift) & F_MASK];
                        if (k < len) L.ring[(p + k + shift) & F_MASK] = (uint8_t)bv;
                        join_lanes();
                    }
                } else {
                    for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        const uint32_t bv = L.ring[(src + k % dist + shift) & F_MASK];
                        if (k < len) L.ring[(p + k + shift) & F_MASK] = (uint8_t)bv;
                        join_lanes();
                    }
                }
                LZ_COUNT(1);
            } else {
                // dist > len here; sources older than the ring come from HBM
                for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    const uint32_t sp = src + k;
                    const bool useg = k < len && sp + (F_RING - F_MAX_MATCH) <= p;
                    const uint32_t gv = __hip_atomic_load(useg ? ob + sp : safe_byte, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t lv = L.ring[(sp + shift) & F_MASK];
                    if (k < len) L.ring[(p + k + shift) & F_MASK] = (uint8_t)(useg ? gv : lv);
                    join_lanes();
                }
                LZ_COUNT(4);
            }
            lds_sync();
            p += len;
            total += len;
            prev = uni((uint32_t)L.ring[(p - 1 + shift) & F_MASK]);
            mbyte = uni(history_byte_v(p - 1 - rep0));
            LZ_STAMP(5);
        }
        if (pend_len) resolve();
        if (!ok) break;
        total &= 15;
        // eof-mode :never at the declared size (lzma.lisp:644-647): the range decoder must be able to finish;
        // and the chunk must have used exactly its compressed size (lzma2.lisp:200-201)
        if (code != 0 || s.pos() != chunk_end) { ok = false; WHY(5); break; }
    }
    if (!ok) { retry(); return; }
    lds_sync();
    flush(true);

    // ---- xz block end (xz.lisp:426-457) and results ----
    semz_tail t{};
    semz_result r{};
    const uint32_t real_csize = s.pos() - amis;
    uint32_t endoff = s.pos();
    s.slide();
    if (is_xz) {
        if (((un.flags & SEMZ_UF_HAVE_IN_SIZE) && real_csize != un.size_in) ||
            ((un.flags & SEMZ_UF_HAVE_OUT_SIZE) && p != un.size_out)) { retry(); return; }
        const uint32_t pad = (4 - (real_csize & 3)) & 3;
        if (endoff + pad > phys_end) { retry(); return; }
        for (uint32_t k = 0; k < pad; k++)
            if (s.byte() != 0) { retry(); return; }
        endoff += pad;
        const uint32_t ctype = un.param1;
        const uint32_t cs = ctype == SEMZ_CHECK_CRC32 ? 4 : ctype == SEMZ_CHECK_CRC64 ? 8 : ctype == SEMZ_CHECK_SHA256 ? 32 : 0;
        t.check_kind = ctype == SEMZ_CHECK_CRC32 ? SEMZ_TK_CRC32 : ctype == SEMZ_CHECK_CRC64 ? SEMZ_TK_CRC64
                       : ctype == SEMZ_CHECK_SHA256 ? SEMZ_TK_SHA256 : SEMZ_TK_NONE;
        if (endoff + cs > phys_end) { retry(); return; }
        t.have_check = 1;
        const uint8_t *cp = ustart + (endoff - amis);
        for (uint32_t k = 0; k < cs / 4; k++) {
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) v |= (uint32_t)cp[4 * k + i] << (8 * i);
            t.want[k] = v;
        }
        endoff += cs;
    }
    const uint64_t end_bit = 8 * (un.in_off + (uint64_t)(endoff - amis));
    t.trailer_end_bit = end_bit;
    r.status = SEMZ_OK;
    r.detail = 0;                          // high word of the compressed size
    r.aux = real_csize;
    r.in_end_bit = end_bit;
    r.out_len = p;
    r.out_off = un.out_off;
    if (lane == 0) {
        results[u] = r;
        tails[u] = t;
#ifdef SEMZ_STAMPS
        LZ_STAMP(6);
        for (int i = 0; i < 8; i++) atomicAdd(&g_lz_stamps[i], (unsigned long long)t_acc[i]);
        for (int i = 0; i < 6; i++) atomicAdd(&g_lz_stamps[8 + i], (unsigned long long)n_acc[i]);
        atomicAdd(&g_lz_stamps[14], 1ull);
#endif
    }
}

} // namespace

#ifdef SEMZ_STAMPS
extern "C" int semz_debug_lz_stamps(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lz_stamps), sizeof(g_lz_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lz_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

void semz_launch_lzma2_fast(const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits, uint8_t *d_out,
                            size_t out_cap, semz_result *d_results, semz_tail *d_tails, hipStream_t stream)
{
    lzma2_fast_kernel<<<nunits, 64, 0, stream>>>(d_in, in_len, d_units, nunits, d_out, out_cap, d_results, d_tails);
}
