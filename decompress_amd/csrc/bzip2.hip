// bzip2.hip -- gfx950 kernels for bzip2 blocks (one unit = one block).
//
// Replaces, per block, the reference loops
//   parse-bzip2-block-header / selectors / trees   bzip2.lisp:298-395
//   decode-bzip2-huffman+rle2                       bzip2.lisp:222-287
//   decode-bzip2-mtf-in-place                       bzip2.lisp:175-191
//   decode-bzip2-bwt                                bzip2.lisp:137-163
//   derandomize-bzip2-block                         bzip2.lisp:112-121
//   decode-bzip2-rle1 + block CRC                   bzip2.lisp:59-102, 507-524
//
// Kernel sequence (all device-resident, no host round trip):
//   bz_decode_kernel  two waves per block: one parses the header and decodes the
//                     Huffman symbols (64-bit windows, every lane a bit offset)
//                     into an LDS queue, the other expands RLE2 runs and applies
//                     the inverse MTF (list in a VGPR, DPP shifts)
//   bz_sort_kernel    one workgroup per block: byte histogram, prefix sums, stable
//                     counting-sort scatter that builds the merged (next << 8 | byte)
//                     vector
//   bz_walk_kernel    the inverse-BWT walk as up to 4097 concurrent chains per block,
//                     each writing its bytes to a row of its own; persistent waves,
//                     one block per XCD at a time so that its vector is walked in L2
//   bz_link_kernel    one workgroup per block: links the chains in order, copies the
//                     rows to their places, de-randomises, and sizes RLE1
//   bz_offsets_kernel exclusive scan of the block output sizes (SEMZ_OUT_AUTO)
//   bz_rle1_kernel    RLE1 expansion into the final output span
//   checksum_kernel   (checksums.hip) block CRC and its comparison
#include "semz_host.h"

namespace {

constexpr uint32_t BZ_STRIDE = 900032;         // per-block scratch stride (bytes / entries)
// Root table index width of the Huffman trees; longer codes are resolved from limit[].  LDS bounds the number of
// blocks a CU holds (two waves each): 33 KB per block with 11-bit roots (four per CU, 1024 on the chip), 12.5 KB
// with 8-bit roots (twelve per CU).  A batch that fits the chip with the wide tables takes them (fewer long
// codes: 28.4 against 30.4 ms for 600 blocks); a larger one is better off with more blocks in flight
// (4800 blocks: 135 ms with 11 bits, 96 with 10, 86 with 9, 79 with 8).
constexpr int BZ_ROOT_WIDE = 11, BZ_ROOT_NARROW = 8;
constexpr uint32_t BZ_WIDE_BLOCKS = 1024;      // batches up to this many blocks use the wide tables
constexpr int BZ_QCAP = 1024;                  // LDS symbol queue between the Huffman wave and the MTF wave
constexpr int BZ_IN_HALF = 1024;               // input is staged in aligned 1 KiB pieces ...
constexpr int BZ_IN_RING = 2 * BZ_IN_HALF;     // ... into a two-piece LDS ring
constexpr int BZ_GROUP = 50;                   // symbols per selector (bzip2.lisp:222-287)
constexpr int BZ_AB_BOUND = 19;                // +bzip2-ab-bound+, bzip2.lisp:217-219
constexpr int BZ_SEGS = 1024;                  // RLE1 segments per block (= threads of bz_bwt_kernel)
constexpr int BZ_CHAINS = 16384;               // inverse-BWT chain starts per block: markers every S = 64 positions at 900 k
// A chain's bytes go to a row of its own, 4 S bytes long; a chain that is longer goes on in a fresh row under a
// fresh id: at most n / S + 2 + n / (4 S) + 2 ids, 5 n bytes and a little.
constexpr int BZ_IDS = 20608;                  // 16386 markers + 4098 continuations at the most, and some
constexpr int BZ_SUPER = 16;                   // bz_link_kernel: every 16th id heads a run of rows
constexpr size_t BZ_TEMP_STRIDE = 5 * (size_t)900032 + 65536;
constexpr int LINK_DEPTH = 16;                 // bz_link_kernel: rows a wave has in flight while it copies
constexpr int BWT_THREADS = 512;               // bz_sort_kernel, bz_link_kernel: one workgroup per block
constexpr int BWT_WAVES = BWT_THREADS / 64;
// bz_walk_kernel: the chip's eight XCDs (one L2 each) take the blocks in turn, so that the one vector an XCD is
// walking stays in its L2 (scripts/ubench/l2_gather.hip: 3.6 MB gathered at random by one XCD's workgroups stay
// resident, 245 ns per dependent read; 5 MB do not).
constexpr int BZ_XCDS = 8;
constexpr int WALK_WGS = 64;                   // workgroups per XCD: 8192 lanes, one wave on every SIMD
constexpr int WALK_THREADS = 128;
constexpr int WALK_STEPS = 4;                  // steps between two refills of a wave's idle lanes
constexpr int WALK_KEEP = 24;                   // a wave goes on to its XCD's next block when this few of its lanes still walk the old one
constexpr int WALK_GRAB = 32;                 // chain starts a wave takes per visit to a block's counter

struct bz_row {            // one row of the walk: how many bytes it holds and which row follows
    uint32_t len, next;
};
// (streaming stores: what the walk writes is not read again by it and must not push the vector out of L2)
__device__ __forceinline__ void put_row(bz_row *r, uint32_t len, uint32_t next)
{
    __builtin_nontemporal_store(((uint64_t)next << 32) | len, reinterpret_cast<uint64_t *>(r));
}

#ifdef SEMZ_STAMPS
// diagnostic build only (make STAMPS=1): shader-clock totals of bz_decode_kernel summed over all blocks
__device__ unsigned long long g_bz_stamps[16];
#define BZ_STAMP_DECL uint64_t t_last = __builtin_amdgcn_s_memtime(), t_acc[5] = {0, 0, 0, 0, 0}; uint32_t n_acc[6] = {0, 0, 0, 0, 0, 0};
#define BZ_STAMP(i) do { uint64_t t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#define BZ_COUNT(i) (n_acc[i]++)
#else
#define BZ_STAMP_DECL
#define BZ_STAMP(i) do { } while (0)
#define BZ_COUNT(i) do { } while (0)
#endif

struct bz_meta {
    uint32_t n;            // bytes entering the inverse BWT
    uint32_t orig;
    uint32_t randomized;
    uint32_t want_crc;
    int32_t status;
    uint32_t detail;
    uint64_t end_bit;      // bit offset inside `in` after the block's end-of-block symbol
    uint64_t out_len;      // bytes after RLE1
    uint64_t out_off;
};

struct bz_seg {            // RLE1 entry state of one segment (bz_bwt_kernel -> bz_rle1_kernel)
    uint32_t off;          // output offset of the segment inside the block's output
    int16_t last;          // -1 = none
    int16_t reps;
};

// bzip2's randtable.c (tables.lisp:85-149)
__constant__ uint16_t BZ_RAND[512] = {
 619,720,127,481,931,816,813,233,566,247,985,724,205,454,863,491,741,242,949,214,
 733,859,335,708,621,574,73,654,730,472,419,436,278,496,867,210,399,680,480,51,
 878,465,811,169,869,675,611,697,867,561,862,687,507,283,482,129,807,591,733,623,
 150,238,59,379,684,877,625,169,643,105,170,607,520,932,727,476,693,425,174,647,
 73,122,335,530,442,853,695,249,445,515,909,545,703,919,874,474,882,500,594,612,
 641,801,220,162,819,984,589,513,495,799,161,604,958,533,221,400,386,867,600,782,
 382,596,414,171,516,375,682,485,911,276,98,553,163,354,666,933,424,341,533,870,
 227,730,475,186,263,647,537,686,600,224,469,68,770,919,190,373,294,822,808,206,
 184,943,795,384,383,461,404,758,839,887,715,67,618,276,204,918,873,777,604,560,
 951,160,578,722,79,804,96,409,713,940,652,934,970,447,318,353,859,672,112,785,
 645,863,803,350,139,93,354,99,820,908,609,772,154,274,580,184,79,626,630,742,
 653,282,762,623,680,81,927,626,789,125,411,521,938,300,821,78,343,175,128,250,
 170,774,972,275,999,639,495,78,352,126,857,956,358,619,580,124,737,594,701,612,
 669,112,134,694,363,992,809,743,168,974,944,375,748,52,600,747,642,182,862,81,
 344,805,988,739,511,655,814,334,249,515,897,955,664,981,649,113,974,459,893,228,
 433,837,553,268,926,240,102,654,459,51,686,754,806,760,493,403,415,394,687,700,
 946,670,656,610,738,392,760,799,887,653,978,321,576,617,626,502,894,679,243,440,
 680,879,194,572,640,724,926,56,204,700,707,151,457,449,797,195,791,558,945,679,
 297,59,87,824,713,663,412,693,342,606,134,108,571,364,631,212,174,643,304,329,
 343,97,430,751,497,314,983,374,822,928,140,206,73,263,980,736,876,478,430,305,
 170,514,364,692,829,82,855,953,676,246,369,970,294,750,807,827,150,790,288,923,
 804,378,215,828,592,281,565,555,710,82,896,831,547,261,524,462,293,465,502,56,
 661,821,976,991,658,869,905,758,745,193,768,550,608,933,378,286,215,979,792,961,
 61,688,793,644,986,403,106,366,905,644,372,567,466,434,645,210,389,550,919,135,
 780,773,635,389,707,100,626,958,165,504,920,176,193,713,857,265,203,50,668,108,
 645,990,626,197,510,357,358,850,858,364,936,638 };

template <int BZ_ROOT_BITS> struct BzTree {
    uint16_t root[1 << BZ_ROOT_BITS];   // (symbol << 5) | length, 0 = code longer than the root
    uint16_t sorted[260];
    uint32_t limit[21];                  // (first+count) << (20-len)
    uint32_t first[21];
    uint16_t offs[21];
    uint32_t maxlen;
};

struct BzShare {           // producer/consumer hand-over (LDS, workgroup-scope atomics on the flags)
    uint32_t produced;     // symbols appended to symq so far (producer)
    uint32_t consumed;     // symbols taken out of it (consumer)
    uint32_t done;         // the producer has ended: status .. end_hi are final
    uint32_t abort;        // the consumer has failed: the producer may stop
    uint32_t hdr_ready;    // header parsed: L.symbols holds the MTF alphabet
    int32_t status;
    uint32_t detail, orig, randomized, want_crc, end_lo, end_hi;
};

template <int BZ_ROOT_BITS> struct __align__(16) BzLds {
    uint8_t inbuf[BZ_IN_RING + 16];      // the first 16 bytes again behind the end: a window's second dword never wraps
    BzTree<BZ_ROOT_BITS> tree[6];
    uint8_t lens[260];
    uint8_t symbols[256];
    uint16_t count[32];
    uint16_t symq[BZ_QCAP];              // decoded symbols waiting for the MTF wave (ring)
    uint32_t mtfq[64];                   // the MTF symbols of the consumer's batch, run symbols squeezed out
    int32_t status;
    BzShare sh;
};

// inclusive prefix sum over the wave
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}

// sum over the wave, the same value in every lane's scalar view
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Lists of up to six trees, one per nibble: the list `a` rearranged as `b` says (nibble j of the result is the
// nibble of `a` that nibble j of `b` names).  Rearranging by b and then by c is rearranging by bz_list_after(b, c).
__device__ __forceinline__ uint32_t bz_list_after(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) r |= ((a >> (4 * ((b >> (4 * j)) & 7))) & 15) << (4 * j);
    return r;
}

// MSB-first bit reader (msb-bit-reader, bits.lisp:121).  Every lane of the
// wave holds the same state.  The input is staged cooperatively (64 lanes x 16
// bytes) in aligned 1 KiB pieces into a 2 KiB LDS ring, so the piece before the
// one staged last stays readable.
struct MsbReader {
    const uint8_t *in_v0;   // in_v0 + v = address of v-offset v (see inflate.hip)
    uint8_t *ring;
    uint64_t in_lo, limit;  // valid v-offsets are [in_lo, limit)
    uint64_t wpos;          // v-offset of the next dword
    uint64_t s_hi;          // staged: [s_hi - BZ_IN_RING, s_hi) (what lies outside [in_lo, limit) reads as zero)
    uint64_t buf;           // bits left aligned: next bit is bit 63
    int cnt;
    bool near_end;

    __device__ __forceinline__ void stage()
    {
        // all 64 lanes: load the piece [s_hi, s_hi + 1 KiB)
        const int lane = threadIdx.x & 63;
        const uint64_t v = s_hi + 16u * lane;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (v >= in_lo && v + 16 <= limit) x = *reinterpret_cast<const uint4 *>(in_v0 + v);
        else if (v < limit && v + 16 > in_lo) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int b = 0; b < 16; b++) {
                uint64_t vv = v + b;
                uint32_t byte = (vv >= in_lo && vv < limit) ? in_v0[vv] : 0u;
                w[0] = (w[0] >> 8) | (w[1] << 24);       // 128-bit shift right by one byte, new byte on top:
                w[1] = (w[1] >> 8) | (w[2] << 24);       // after 16 steps byte b sits at position b
                w[2] = (w[2] >> 8) | (w[3] << 24);
                w[3] = (w[3] >> 8) | (byte << 24);
            }
            x = make_uint4(w[0], w[1], w[2], w[3]);
        }
        lds_sync();
        *reinterpret_cast<uint4 *>(ring + ((uint32_t)s_hi & (BZ_IN_RING - 1)) + 16 * lane) = x;
        if (((uint32_t)s_hi & (BZ_IN_RING - 1)) == 0 && lane == 0) *reinterpret_cast<uint4 *>(ring + BZ_IN_RING) = x;   // (the mirror)
        __builtin_amdgcn_wave_barrier();
        lds_sync();
        s_hi += BZ_IN_HALF;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            if (wpos >= s_hi) stage();
            uint32_t w = *reinterpret_cast<const uint32_t *>(ring + ((uint32_t)wpos & (BZ_IN_RING - 1)));
            w = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bswap32(w));
            if (wpos + 4 > limit) near_end = true;
            wpos += 4;
            buf |= (uint64_t)w << (32 - cnt);
            cnt += 32;
        }
    }
    __device__ __forceinline__ void init(const uint8_t *v0, uint8_t *lds, uint64_t lo, uint64_t lim, uint64_t v, uint32_t bit_off)
    {
        in_v0 = v0; ring = lds; in_lo = lo; limit = lim;
        wpos = v & ~3ull; buf = 0; cnt = 0; near_end = false;
        // position of the first bit inside the first dword
        int skip = (int)(v & 3) * 8 + (int)bit_off;
        s_hi = wpos & ~(uint64_t)(BZ_IN_HALF - 1);
        stage();
        refill();
        buf <<= skip;
        cnt -= skip;
    }
    __device__ __forceinline__ int64_t remaining() const { return (int64_t)(8 * limit) - (int64_t)(8 * wpos) + cnt; }
    __device__ __forceinline__ bool short_of(int n) const { return near_end && remaining() < n; }
    __device__ __forceinline__ uint64_t bitpos() const { return 8 * wpos - (uint64_t)cnt; }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(buf >> (64 - n)); }
    __device__ __forceinline__ void drop(int n) { buf <<= n; cnt -= n; }
    // read n <= 24 bits; sets eof instead of reading past the end
    __device__ __forceinline__ uint32_t read(int n, bool &eof)
    {
        refill();
        if (short_of(n)) { eof = true; return 0; }
        uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

// One canonical tree from lens[0..n) (lengths->ht :be, huffman.lisp:142-225).
template <int BZ_ROOT_BITS> __device__ int bz_build_tree(BzLds<BZ_ROOT_BITS> &L, BzTree<BZ_ROOT_BITS> &T, int n)
{
    const int lane = threadIdx.x & 63;
    uint32_t c = 0;
    if (lane >= 1 && lane <= 20)
        for (int s = 0; s < n; s++) c += (L.lens[s] == lane);
    if (lane < 32) L.count[lane] = (uint16_t)c;
    lds_sync();
    if (lane == 0) {
        uint32_t code = 0, off = 0, total = 0, maxl = 1;
        T.first[0] = 0; T.offs[0] = 0; T.limit[0] = 0;
        for (int l = 1; l <= 20; l++) {
            uint32_t cnt = L.count[l];
            code <<= 1;
            T.first[l] = code;
            T.offs[l] = (uint16_t)off;
            code += cnt;
            off += cnt;
            T.limit[l] = code << (20 - l);
            total += cnt << (20 - l);
            if (cnt) maxl = l;
        }
        T.maxlen = maxl;
        L.status = total < (1u << 20) ? SEMZ_E_HUFF_UNDERFULL : total > (1u << 20) ? SEMZ_E_HUFF_OVERFULL : SEMZ_OK;
    }
    lds_sync();
    int st = L.status;
    if (st != SEMZ_OK) return st;
    if (lane >= 1 && lane <= 20) {
        uint32_t k = T.offs[lane];
        for (int s = 0; s < n; s++)
            if (L.lens[s] == lane) T.sorted[k++] = (uint16_t)s;
    }
    lds_sync();
    for (int i = lane; i < (1 << BZ_ROOT_BITS); i += 64) {
        uint32_t v20 = (uint32_t)i << (20 - BZ_ROOT_BITS);
        uint16_t e = 0;
        for (int l = 1; l <= BZ_ROOT_BITS; l++) {
            if (v20 < T.limit[l]) {
                uint32_t idx = T.offs[l] + (v20 >> (20 - l)) - T.first[l];
                e = (uint16_t)((T.sorted[idx] << 5) | l);
                break;
            }
        }
        T.root[i] = e;
    }
    lds_sync();
    return SEMZ_OK;
}

template <int BZ_ROOT_BITS>
__global__ __launch_bounds__(128) void bz_decode_kernel(const uint8_t *in, uint64_t in_len, const semz_unit *units, uint32_t nunits,
                                                        uint64_t out_total, bz_meta *meta, uint8_t *pre_all, uint8_t *post_all)
{
    // Two waves per block, a producer and a consumer, joined by a symbol queue in LDS.  A lone wave issues one
    // instruction every ~5 clocks whatever its kind and a bzip2 batch has fewer blocks than the chip has SIMDs,
    // so the two halves of the work run side by side and each is organised to spend few instructions per symbol.
    //
    // Wave 0: header, selectors, trees, then Huffman.  Lane l looks up the code that would start at bit P + l,
    // so one pair of LDS round trips (input window, root table) serves every symbol that starts inside the next
    // 64 bits.  The wave follows the chain off -> off + len(off) with readlane and collects the start positions
    // in a 64-bit mask; group boundary (50 symbols per selector) and end-of-block are applied to the mask
    // afterwards, and the surviving lanes append their symbols to the queue in order (rank = mbcnt of the mask).
    //
    // Wave 1: RLE2 + MTF, 64 queued symbols at a time.  Every symbol is RUNA/RUNB or an MTF index, so the run
    // symbols between two MTF symbols are one run whose bits come straight out of the RUNB ballot; only MTF
    // symbols are walked one by one.  The MTF list lives in four VGPRs (one entry per lane): an index below 64
    // (nearly all of them, on text) is a readlane, a DPP wave shift and a select in the first of them; a larger
    // one shifts whole registers with a carry lane.  The bytes of consecutive MTF symbols collect in one VGPR (lane
    // e = e-th of them) and are stored together in front of the next run or at the end of the batch; a run is
    // stored by all lanes at once (nothing in this kernel reads `pre` back).
    //
    // Failures: the consumer's own failure comes first in stream order and wins; a failure of the producer is
    // reported once the consumer has drained the queue.  Both waves always reach their end: the producer sets
    // `done` on every path and leaves its wait loop when the consumer sets `abort`; the consumer only waits for
    // symbols or `done`.
    __shared__ BzLds<BZ_ROOT_BITS> L;
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < sizeof(BzShare) / 4) reinterpret_cast<uint32_t *>(&L.sh)[threadIdx.x] = 0;
    __syncthreads();
    const semz_unit un = units[u];
    if (!semz_unit_spans_ok(un, in_len, out_total, 1)) {         // nothing of this unit is touched
        if (threadIdx.x == 0) {
            bz_meta m{};
            m.status = SEMZ_E_BAD_UNIT;
            meta[u] = m;
        }
        return;
    }
    uint8_t *pre = pre_all + (size_t)u * BZ_STRIDE;
    // The selectors (four bits each, up to 2^15 - 1 of them: 16 KiB) wait in the block's `post` array, which
    // nothing else uses before bz_link_kernel, and come back 512 at a time into one vector register: LDS is what
    // bounds the number of blocks a CU holds, and a batch of config 3's size has six times as many blocks as fit.
    uint32_t *selg = reinterpret_cast<uint32_t *>(post_all + (size_t)u * BZ_STRIDE);
    const uint32_t level = un.param0 >= 1 && un.param0 <= 9 ? un.param0 : 9;
    const uint32_t block_size = level * 100000u;
    const uint64_t amis = (uint64_t)((uintptr_t)in & 15);
    BZ_STAMP_DECL
#define BZ_LD(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&(x), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)))
#define BZ_ST(x, v) __hip_atomic_store(&(x), (uint32_t)(v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)

    if (wid == 0) {
        // ================= producer =================
        MsbReader br;
        br.init(in - amis, L.inbuf, amis, un.in_limit + amis, un.in_off + amis, un.in_bit_off);
        int status = SEMZ_OK;
        uint32_t detail = 0;
        bool eof = false;
        uint32_t orig = 0, randomized = 0, want_crc = 0;
        uint64_t P = 0;                 // bit position (v-space) once the symbol loop has taken over from `br`
        bool in_symbols = false;
#define BZ_FAIL(code, det) do { status = (code); detail = (det); goto done; } while (0)
#define BZ_EOFCHK() do { if (eof) BZ_FAIL(SEMZ_EOF, 0); } while (0)
        {
            // 48-bit block magic (bzip2.lisp:460-463)
            uint32_t m1 = br.read(24, eof); BZ_EOFCHK();
            uint32_t m2 = br.read(24, eof); BZ_EOFCHK();
            if (m1 != 0x314159u || m2 != 0x265359u) BZ_FAIL(SEMZ_E_BZ_BLOCK_MAGIC, 0);
            // parse-bzip2-block-header, bzip2.lisp:351-395
            uint32_t c1 = br.read(16, eof); BZ_EOFCHK();
            uint32_t c2 = br.read(16, eof); BZ_EOFCHK();
            want_crc = (c1 << 16) | c2;
            randomized = br.read(1, eof); BZ_EOFCHK();
            orig = br.read(24, eof); BZ_EOFCHK();
            uint32_t l1 = br.read(16, eof); BZ_EOFCHK();
            if (l1 == 0) BZ_FAIL(SEMZ_E_BZ_NO_SYMBOLS, 0);
            uint32_t nsym = 0;
            for (int i = 0; i < 16; i++) {
                if (!((l1 >> (15 - i)) & 1)) continue;
                uint32_t l2 = br.read(16, eof); BZ_EOFCHK();
                for (int j = 0; j < 16; j++)
                    if ((l2 >> (15 - j)) & 1) {
                        if (lane == 0) L.symbols[nsym] = (uint8_t)(16 * i + j);
                        nsym++;
                    }
            }
            uint32_t ntrees = br.read(3, eof); BZ_EOFCHK();
            uint32_t nsel = br.read(15, eof); BZ_EOFCHK();
            if (ntrees < 2 || ntrees > 6) BZ_FAIL(SEMZ_E_BZ_TREE_COUNT, ntrees);
            if (nsel == 0) BZ_FAIL(SEMZ_E_BZ_NO_SELECTORS, 0);
            // read-bzip2-selectors (bzip2.lisp:302-311): codes 0,10,110,1110,11110,111110,111111, then MTF
            {
                uint32_t stack = 0x543210u;                 // nibble j = tree at MTF position j
                uint32_t ndone = 0;
                // Every code of a tree the block has ends in a 0, so selector k ends at the k-th 0 bit and its
                // MTF index is the number of ones in front of it: 2048 bits at a time, lane l takes the zeros of
                // bits [32 l, 32 l + 32).  Moving to the front rearranges the list, and rearrangements compose:
                // each lane walks its own zeros from the list 0..5 (pass 1: which place of the list it was handed
                // each selector takes, and the lane's whole rearrangement), a scan over the lanes gives each lane
                // the list it is handed, and a second, short walk looks the trees up in it.  Anything that may
                // fail (a run of ones that is no tree of this block, the end of the input within reach) is left
                // to the loop below, which starts where the last good piece ended.
                uint8_t *stg = reinterpret_cast<uint8_t *>(L.tree);    // (the trees are built after this)
                static_assert(sizeof(L.tree) >= 2048 + 8, "selector staging");
                uint64_t B = br.bitpos();
                const uint64_t B0 = B;
                while (ndone < nsel) {
                    if (8 * br.limit < B + 2048 + 64 || (B >> 5) * 4 < br.in_lo) break;
                    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(br.in_v0) + (B >> 5);
                    const uint32_t wa = __builtin_bswap32(w32[lane]), wb = __builtin_bswap32(w32[lane + 1]);
                    const uint32_t z = ~(uint32_t)((((((uint64_t)wa << 32) | wb) << ((uint32_t)B & 31))) >> 32);
                    const uint32_t n = (uint32_t)__builtin_popcount(z);
                    const uint32_t incl = wave_scan_add(n), basel = incl - n;
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (total == 0) break;
                    const uint32_t used = total < nsel - ndone ? total : nsel - ndone;
                    const uint32_t neff = basel < used ? (n < used - basel ? n : used - basel) : 0u;
                    // ones at the end of the lane below: the front of this lane's first selector
                    const uint32_t zprev = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)z, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    uint32_t c = zprev ? (uint32_t)__builtin_ctz(zprev) : 32u;
                    uint32_t zs = z, S = 0x543210u;
                    bool bad = false;
                    const uint32_t off = (ndone & 7) + basel;
                    for (uint32_t it = 0; __builtin_amdgcn_ballot_w64(it < neff); it++) {
                        const bool act = it < neff;
                        const uint32_t tz = zs ? (uint32_t)__builtin_clz(zs) : 0u;
                        uint32_t idx = tz + c;
                        c = 0;
                        zs = (zs << tz) << 1;
                        bad |= act && idx >= ntrees;
                        if (idx > 5) idx = 5;
                        const uint32_t sh = 4 * idx, val = (S >> sh) & 15;
                        const uint32_t S2 = (S & ~((1u << (sh + 4)) - 1)) | ((S & ((1u << sh) - 1)) << 4) | val;
                        if (act) {
                            S = S2;
                            stg[off + it] = (uint8_t)val;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(bad)) break;
                    uint32_t X = S;
#pragma unroll
                    for (int d = 1; d < 64; d *= 2) {
                        const uint32_t below = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane - d), (int)X);
                        if (lane >= d) X = bz_list_after(below, X);
                    }
                    const uint32_t xbelow = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane - 1), (int)X);
                    const uint32_t handed = lane ? bz_list_after(stack, xbelow) : stack;
                    stack = bz_list_after(stack, (uint32_t)__builtin_amdgcn_readlane((int)X, 63));
                    for (uint32_t it = 0; __builtin_amdgcn_ballot_w64(it < neff); it++)
                        if (it < neff) stg[off + it] = (uint8_t)((handed >> (4 * stg[off + it])) & 15);
                    lds_sync();
                    const uint32_t cnt = (ndone & 7) + used;
                    const bool last = ndone + used == nsel;
                    const uint32_t nwords = last ? (cnt + 7) >> 3 : cnt >> 3;
                    for (uint32_t m = (uint32_t)lane; m < nwords; m += 64) {
                        uint64_t x = *reinterpret_cast<const uint64_t *>(stg + 8 * m);
                        if (cnt - 8 * m < 8) x &= (1ull << (8 * (cnt - 8 * m))) - 1;
                        uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
                        lo = (lo | (lo >> 4)) & 0x00FF00FFu; lo = (lo | (lo >> 8)) & 0xFFFFu;
                        hi = (hi | (hi >> 4)) & 0x00FF00FFu; hi = (hi | (hi >> 8)) & 0xFFFFu;
                        selg[(ndone >> 3) + m] = lo | (hi << 16);
                    }
                    lds_sync();
                    const uint32_t rem = last ? 0u : cnt & 7;
                    const uint8_t keep = (uint32_t)lane < rem ? stg[(cnt & ~7u) + lane] : (uint8_t)0;
                    lds_sync();
                    if ((uint32_t)lane < rem) stg[lane] = keep;
                    lds_sync();
                    // the last 0 this piece used
                    const uint32_t src = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(basel < used && used <= incl));
                    uint32_t zz = (uint32_t)__builtin_amdgcn_readlane((int)z, (int)src);
                    const uint32_t skip = used - (uint32_t)__builtin_amdgcn_readlane((int)basel, (int)src) - 1;
                    for (uint32_t i = 0; i < skip; i++) zz &= ~(0x80000000u >> __builtin_clz(zz));
                    B += 32 * src + (uint32_t)__builtin_clz(zz) + 1;
                    ndone += used;
                }
                uint32_t packed = 0;                        // eight selectors per word
                if (B != B0) {
                    br.init(br.in_v0, br.ring, br.in_lo, br.limit, B >> 3, (uint32_t)(B & 7));
                    for (uint32_t i = 0; i < (ndone & 7) && ndone < nsel; i++)
                        packed |= (uint32_t)__builtin_amdgcn_readfirstlane((int)stg[i]) << (4 * i);
                }
                for (uint32_t i = ndone; i < nsel; i++) {
                    br.refill();
                    if (br.short_of(6)) BZ_FAIL(SEMZ_EOF, 0);
                    uint32_t six = br.peek(6);
                    uint32_t idx = (uint32_t)__builtin_clz(~(six << 26));   // leading ones
                    if (idx > 6) idx = 6;
                    br.drop(idx < 6 ? idx + 1 : 6);
                    if (idx >= ntrees) BZ_FAIL(SEMZ_E_BZ_SELECTOR, idx);
                    uint32_t val = (stack >> (4 * idx)) & 15;
                    uint32_t low = stack & ((1u << (4 * idx)) - 1);
                    uint32_t high = idx >= 7 ? 0 : (stack >> (4 * (idx + 1))) << (4 * (idx + 1));
                    stack = high | (low << 4) | val;
                    packed |= val << (4 * (i & 7));
                    if ((i & 7) == 7 || i + 1 == nsel) {
                        if (lane == 0) selg[i >> 3] = packed;
                        packed = 0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // (written by this wave, read back below)
            }
            // read-bzip2-trees (bzip2.lisp:317-343): 5-bit start, then 0 = emit, 10 = +1, 11 = -1
            for (uint32_t t = 0; t < ntrees; t++) {
                int cur = (int)br.read(5, eof); BZ_EOFCHK();
                for (uint32_t i = 0; i < nsym + 2; i++) {
                    for (;;) {
                        if (cur < 1 || cur > 20) BZ_FAIL(SEMZ_E_BZ_CODELEN, 0);
                        br.refill();
                        if (br.short_of(2)) BZ_FAIL(SEMZ_EOF, 0);
                        uint32_t two = br.peek(2);
                        if (two < 2) { br.drop(1); break; }
                        br.drop(2);
                        cur += two == 2 ? 1 : -1;
                    }
                    if (lane == 0) L.lens[i] = (uint8_t)cur;
                }
                lds_sync();
                // a call result counts as divergent: say that it is uniform, or every value after this loop lands in VGPRs
                int ts = __builtin_amdgcn_readfirstlane(bz_build_tree(L, L.tree[t], (int)nsym + 2));
                if (ts != SEMZ_OK) BZ_FAIL(ts, 0);
            }
            if (nsym == 0) BZ_FAIL(SEMZ_E_BZ_NO_SYMBOLS, 1);
            lds_sync();
            BZ_ST(L.sh.hdr_ready, 1);

            const uint32_t eob = nsym + 1;
            const uint64_t endbits = 8 * br.limit;
            const uint32_t *ring32 = reinterpret_cast<const uint32_t *>(L.inbuf);
            uint32_t produced = 0, consumed_seen = 0;
            uint32_t g = 0, k = BZ_GROUP, t = 0;
            uint32_t selw = 0, selblk = ~0u;                // lane j: the 8 selectors 512 selblk + 8 j ..
            P = br.bitpos();
            in_symbols = true;
            // (the block of assembler below writes LDS addresses as immediates: this kernel's only LDS object starts at 0)
            const bool lds_at_0 = (uint32_t)__builtin_amdgcn_readfirstlane(
                                      (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)L.inbuf) == 0;
            BZ_STAMP(0);
            for (;;) {
                // room for one more window of symbols?
                if (produced - consumed_seen > (uint32_t)BZ_QCAP - 64) {
                    bool stop = false;
                    for (;;) {
                        consumed_seen = BZ_LD(L.sh.consumed);
                        if (produced - consumed_seen <= (uint32_t)BZ_QCAP - 64) break;
                        if (BZ_LD(L.sh.abort)) { stop = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    BZ_STAMP(3);
                    if (stop) break;
                }
                if (k == (uint32_t)BZ_GROUP) {
                    if (g >= nsel) BZ_FAIL(SEMZ_E_BZ_SELECTORS_EXHAUSTED, 0);
                    if ((g & 31) == 0) semz_take_turn(15);               // (semz_device.h; every 1600 symbols)
                    if ((g >> 9) != selblk) {
                        selblk = g >> 9;
                        selw = __hip_atomic_load(selg + 64 * selblk + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    t = ((uint32_t)__builtin_amdgcn_readlane((int)selw, (int)((g >> 3) & 63)) >> (4 * (g & 7))) & 15;
                    g++;
                    k = 0;
                }
                const BzTree<BZ_ROOT_BITS> &T = L.tree[t];
                while ((P >> 3) + 16 > br.s_hi) { br.stage(); BZ_COUNT(3); }
                // ---- runs of ordinary windows, one block of assembler ----
                // Nearly every window has nothing to it: every code of the chain fits the root table, the block does
                // not end, input is staged and the queue has room.  Written in C++ such a window is ~100 instructions
                // and nine taken branches around its 15 chain steps (the compiler keeps every condition as a flag
                // register and branches on it twice); here it is 50 and one.  The block takes the step from one group
                // to the next (the selector's tree) and cuts a window at the end of a group by itself; it commits
                // what it takes of a window at once and leaves in front of the first window it does not take -- a
                // long code or the end of the block in the chain, the queue looking full, the staged input running
                // out, a group at which the wave takes its turn or fetches selectors -- without having touched it:
                // the code below (or the top of the loop) deals with that one as before.
                if (lds_at_0) {
                    const int64_t b1 = (int64_t)(8 * br.s_hi) - 128 - (int64_t)P;            // staged bits ahead
                    const int64_t b2 = (int64_t)(endbits - P) - (64 + 20);                    // bits before the input's last ones
                    const int64_t b = b1 < b2 ? b1 : b2;
                    if (b >= 0) {
                        int32_t bud = b < 0x7FFFFFFF ? (int32_t)b : 0x7FFFFFFF;
                        uint32_t pr = (uint32_t)P;
                        const uint32_t pr0 = pr, produced0 = produced;
                        (void)produced0;
                        constexpr uint32_t TREES = __builtin_offsetof(BzLds<BZ_ROOT_BITS>, tree), TREE_SIZE = sizeof(BzTree<BZ_ROOT_BITS>);
                        uint32_t rootb = TREES + t * TREE_SIZE;
                        uint32_t x, y, off, cnt, len_s, wins = 0, bp, a0, sh, evv, lenv2, sym, zero;
                        uint64_t longm2, xx;
                        asm volatile(
                            "v_mov_b32_e32 %[zero], 0\n"
                            "10:\n\t"
                            "s_cmp_lt_i32 %[bud], 0\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            "s_sub_u32 %[x], %[prod], %[cons]\n\t"
                            "s_cmp_gt_u32 %[x], %[room]\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            "s_cmp_lt_u32 %[k], %[group]\n\t"
                            "s_cbranch_scc1 11f\n\t"
                            // the next group: its selector is four bits of lane (g >> 3) & 63 of selw
                            "s_cmp_ge_u32 %[g], %[nsel]\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            "s_and_b32 %[x], %[g], 31\n\t"
                            "s_cmp_eq_u32 %[x], 0\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            "s_lshr_b32 %[x], %[g], 9\n\t"
                            "s_cmp_lg_u32 %[x], %[selblk]\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            "s_lshr_b32 %[x], %[g], 3\n\t"
                            "s_lshl_b32 %[y], %[g], 2\n\t"
                            "v_readlane_b32 %[x], %[selw], %[x]\n\t"
                            "s_add_u32 %[g], %[g], 1\n\t"
                            "s_mov_b32 %[k], 0\n\t"
                            "s_lshr_b32 %[x], %[x], %[y]\n\t"
                            "s_and_b32 %[tree], %[x], 15\n\t"
                            "s_mul_i32 %[x], %[tree], %[tsize]\n\t"
                            "s_add_u32 %[rootb], %[x], %[trees]\n"
                            "11:\n\t"
                            // lane l: the 32 bits from bit P + l on, and the root entry of their top bits
                            "v_add_u32_e32 %[bp], %[pr], %[lane]\n\t"
                            "v_lshrrev_b32_e32 %[a0], 3, %[bp]\n\t"
                            "v_and_b32_e32 %[a0], 0x7fc, %[a0]\n\t"
                            "ds_read2_b32 v[40:41], %[a0] offset1:1\n\t"
                            "v_and_b32_e32 %[sh], 31, %[bp]\n\t"
                            "s_mov_b64 s[96:97], 0\n\t"
                            "s_movk_i32 %[off], 0xffc0\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_perm_b32 v43, 0, v40, %[sel]\n\t"
                            "v_perm_b32 v42, 0, v41, %[sel]\n\t"
                            "v_lshlrev_b64 v[42:43], %[sh], v[42:43]\n\t"
                            "v_lshrrev_b32_e32 %[a0], %[rsh], v43\n\t"
                            "v_lshl_add_u32 %[a0], %[a0], 1, %[rootb]\n\t"
                            "ds_read_u16 %[ev], %[a0]\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_and_b32_e32 %[lenv], 31, %[ev]\n\t"
                            "v_cmp_eq_u32_e64 %[longm], 0, %[lenv]\n\t"
                            "v_cndmask_b32_e64 %[lenv], %[lenv], 64, %[longm]\n\t"
                            "v_lshrrev_b32_e32 %[sym], 5, %[ev]\n"
                            // the chain of symbol starts (see below)
                            "1:\n\t"
#define BZ_RUN_STEP "v_readlane_b32 %[t], %[lenv], %[off]\n\ts_bitset1_b64 s[96:97], %[off]\n\ts_add_u32 %[off], %[off], %[t]\n\ts_cbranch_scc1 2f\n\t"
                            BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP BZ_RUN_STEP
#undef BZ_RUN_STEP
                            "s_branch 1b\n"
                            "2:\n\t"
                            // a0 = how many symbols start below the lane
                            "s_bcnt1_i32_b64 %[cnt], s[96:97]\n\t"
                            "v_mbcnt_lo_u32_b32 %[a0], s96, 0\n\t"
                            "s_add_u32 %[x], %[k], %[cnt]\n\t"
                            "v_mbcnt_hi_u32_b32 %[a0], s97, %[a0]\n\t"
                            "s_cmp_gt_u32 %[x], %[group]\n\t"
                            "s_cbranch_scc0 3f\n\t"
                            // the group ends inside the window: the start with `group - k` starts below it is the
                            // next group's first symbol, and the window ends there
                            "s_sub_u32 %[cnt], %[group], %[k]\n\t"
                            "v_cmp_eq_u32_e32 vcc, %[cnt], %[a0]\n\t"
                            "s_and_b64 %[xx], vcc, s[96:97]\n\t"
                            "s_ff1_i32_b64 %[x], %[xx]\n\t"
                            "s_bfm_b64 %[xx], %[x], 0\n\t"
                            "s_sub_u32 %[off], %[x], 64\n\t"
                            "s_and_b64 s[96:97], s[96:97], %[xx]\n\t"
                            "s_mov_b32 %[x], %[group]\n"
                            "3:\n\t"
                            // a long code or the end of the block in what is left of the chain: not this block's
                            "v_cmp_eq_u32_e32 vcc, %[eob], %[sym]\n\t"
                            "s_or_b64 %[xx], vcc, %[longm]\n\t"
                            "s_and_b64 %[xx], %[xx], s[96:97]\n\t"
                            "s_cmp_lg_u64 %[xx], 0\n\t"
                            "s_cbranch_scc1 90f\n\t"
                            // taken: the symbols go to the queue in order, the counters move on
                            "s_mov_b32 %[k], %[x]\n\t"
                            "v_add_u32_e32 %[a0], %[prod], %[a0]\n\t"
                            "v_and_b32_e32 %[a0], %[qmask], %[a0]\n\t"
                            "v_lshlrev_b32_e32 %[a0], 1, %[a0]\n\t"
                            "s_mov_b64 exec, s[96:97]\n\t"
                            "ds_write_b16 %[a0], %[sym] offset:%[symq]\n\t"
                            "s_mov_b64 exec, -1\n\t"
                            "s_add_u32 %[prod], %[prod], %[cnt]\n\t"
                            "s_add_u32 %[off], %[off], 64\n\t"
                            "s_add_u32 %[wins], %[wins], 1\n\t"
                            "s_add_u32 %[pr], %[pr], %[off]\n\t"
                            "s_sub_u32 %[bud], %[bud], %[off]\n\t"
                            "v_mov_b32_e32 %[a0], %[prod]\n\t"
                            // (LDS operations of a wave are carried out in order: the count follows the symbols)
                            "ds_write_b32 %[zero], %[a0] offset:%[prodat]\n\t"
                            "s_branch 10b\n"
                            "90:\n\t"
                            : [pr] "+s"(pr), [bud] "+s"(bud), [k] "+s"(k), [g] "+s"(g), [tree] "+s"(t), [rootb] "+s"(rootb),
                              [prod] "+s"(produced), [wins] "+s"(wins), [x] "=&s"(x), [y] "=&s"(y),
                              [off] "=&s"(off), [cnt] "=&s"(cnt), [t] "=&s"(len_s), [longm] "=&s"(longm2), [xx] "=&s"(xx),
                              [bp] "=&v"(bp), [a0] "=&v"(a0), [sh] "=&v"(sh), [ev] "=&v"(evv), [lenv] "=&v"(lenv2), [sym] "=&v"(sym),
                              [zero] "=&v"(zero)
                            : [cons] "s"(consumed_seen), [eob] "s"(eob), [sel] "s"(0x00010203u), [lane] "v"((uint32_t)lane),
                              [nsel] "s"(nsel), [selblk] "s"(selblk), [selw] "v"(selw),
                              [room] "i"(BZ_QCAP - 64), [group] "i"(BZ_GROUP), [qmask] "i"(BZ_QCAP - 1), [rsh] "i"(32 - BZ_ROOT_BITS),
                              [trees] "i"(TREES), [tsize] "i"(TREE_SIZE),
                              [symq] "i"(__builtin_offsetof(BzLds<BZ_ROOT_BITS>, symq)),
                              [prodat] "i"(__builtin_offsetof(BzLds<BZ_ROOT_BITS>, sh) + __builtin_offsetof(BzShare, produced))
                            : "scc", "vcc", "memory", "v40", "v41", "v42", "v43", "s96", "s97");
                        P += (uint32_t)(pr - pr0);
#ifdef SEMZ_STAMPS
                        n_acc[0] += wins;
                        n_acc[1] += produced - produced0;
#endif
                        BZ_STAMP(1);
                        if (wins) continue;
                    }
                }
                BZ_COUNT(0);
                const uint64_t bp = P + (uint32_t)lane;
                const uint32_t d = (uint32_t)(bp >> 5);
                const uint32_t hi = __builtin_bswap32(ring32[d & (BZ_IN_RING / 4 - 1)]);
                const uint32_t lo = __builtin_bswap32(ring32[(d + 1) & (BZ_IN_RING / 4 - 1)]);
                const uint32_t top = (uint32_t)(((((uint64_t)hi << 32) | lo) << ((uint32_t)bp & 31)) >> 32);
                uint32_t ev = T.root[top >> (32 - BZ_ROOT_BITS)];
                // A code longer than the root index has a zero entry.  Such a lane ends the chain (length 64), and
                // only when the chain gets there -- the lane is a real symbol start, not one of the 60 positions
                // looked up for nothing -- is its code searched for, below.
                const uint64_t longm = __builtin_amdgcn_ballot_w64((ev & 31) == 0);
                // the chain of symbol starts inside this window
                const uint32_t lenv = (ev & 31) ? (ev & 31) : 64u;
                uint64_t mask = 0;
                uint32_t off = 0;
                bool short_input = false;
                if (endbits - P >= 64 + 20) {
                    // Four instructions per symbol, written out because the compiler cannot branch on the carry of
                    // a scalar add: the offset runs biased by 2^32 - 64 (v_readlane and s_bitset1_b64 only look at
                    // its low six bits), so the add carries out exactly when the chain leaves the window.  Every
                    // lane holds a length of at least 1: the trees are complete (bz_build_tree).  The s_nop is the
                    // wait state gfx950 wants between the VALU write of lenv and a v_readlane of it.
                    uint32_t offb = 0u - 64u, len;
                    // (eight steps written out between two jumps back: a taken branch costs three times what a
                    // branch that falls through does, and this wave has its SIMD to itself in a batch of 600 blocks)
#define BZ_CHAIN_STEP "v_readlane_b32 %[t], %[tb], %[off]\n\ts_bitset1_b64 %[m], %[off]\n\ts_add_u32 %[off], %[off], %[t]\n\ts_cbranch_scc1 2f\n\t"
                    asm volatile("s_nop 0\n"
                                 "1:\n\t"
                                 BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP BZ_CHAIN_STEP
                                 "s_branch 1b\n"
                                 "2:"
                                 : [m] "+s"(mask), [off] "+s"(offb), [t] "=&s"(len)
                                 : [tb] "v"(lenv)
                                 : "scc");
#undef BZ_CHAIN_STEP
                    off = offb + 64u;
                } else {
                    // the last bits of the input: the reference runs out of data when fewer bits than the
                    // longest code of the tree are left (trees are always full-read, huffman.lisp)
                    const uint32_t maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.maxlen);
                    do {
                        if (endbits - (P + off) < maxlen) { short_input = true; break; }
                        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)lenv, (int)off);
                        mask |= 1ull << off;
                        off += len;
                    } while (off < 64);
                }
                if (longm & mask) {
                    // the chain's last symbol has a long code: lane i tries length BZ_ROOT_BITS + 1 + i, the
                    // shortest length whose limit lies above the code wins (complete trees: one does)
                    BZ_COUNT(2);
                    const uint32_t j = (uint32_t)__builtin_ctzll(longm & mask);
                    const uint32_t v20 = (uint32_t)__builtin_amdgcn_readlane((int)top, (int)j) >> 12;
                    const uint32_t l = (uint32_t)lane < 20u - BZ_ROOT_BITS ? (uint32_t)BZ_ROOT_BITS + 1 + (uint32_t)lane : 20u;
                    const uint32_t lim = T.limit[l], first = T.first[l], offs = T.offs[l];
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(v20 < lim);
                    const uint32_t w = hit ? (uint32_t)__builtin_ctzll(hit) : 0u;
                    const uint32_t idx = offs + (v20 >> (20 - l)) - first;       // (only the winning lane's is meaningful)
                    const uint32_t cand = ((uint32_t)T.sorted[idx < 260u ? idx : 0u] << 5) | l;
                    const uint32_t found = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)w);
                    if ((uint32_t)lane == j) ev = found;
                    off = j + (found & 31);
                }
                // group boundary: symbols past it were looked up in the wrong tree
                uint32_t cnt = (uint32_t)__builtin_popcountll(mask);
                if (cnt > (uint32_t)BZ_GROUP - k) {
                    // the start with BZ_GROUP - k starts below it is the first symbol of the next group
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                    const uint64_t cut = __builtin_amdgcn_ballot_w64(below == (uint32_t)BZ_GROUP - k) & mask;
                    off = (uint32_t)__builtin_ctzll(cut);
                    mask &= (1ull << off) - 1;
                    cnt = (uint32_t)BZ_GROUP - k;
                    short_input = false;
                }
                // end of block
                const uint64_t eobm = __builtin_amdgcn_ballot_w64((ev >> 5) == eob) & mask;
                uint64_t qmask = mask;
                bool finished = false;
                if (eobm) {
                    const uint32_t e = (uint32_t)__builtin_ctzll(eobm);
                    qmask = mask & ((1ull << e) - 1);
                    off = e + ((uint32_t)__builtin_amdgcn_readlane((int)ev, (int)e) & 31);
                    finished = true;
                    short_input = false;
                }
                // append the symbols to the queue in order and publish them
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(qmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qmask, 0));
                if (__builtin_amdgcn_inverse_ballot_w64(qmask)) L.symq[(produced + rank) & (BZ_QCAP - 1)] = (uint16_t)(ev >> 5);
                produced += (uint32_t)__builtin_popcountll(qmask);
                k += cnt;
                P += off;
                BZ_ST(L.sh.produced, produced);
#ifdef SEMZ_STAMPS
                n_acc[1] += cnt;
#endif
                BZ_STAMP(1);
                if (short_input) BZ_FAIL(SEMZ_EOF, 0);
                if (finished) break;
            }
        }
    done:
        L.sh.status = status;
        L.sh.detail = detail;
        L.sh.orig = orig;
        L.sh.randomized = randomized;
        L.sh.want_crc = want_crc;
        {
            const uint64_t end_bit = (in_symbols ? P : br.bitpos()) - 8 * amis;
            L.sh.end_lo = (uint32_t)end_bit;
            L.sh.end_hi = (uint32_t)(end_bit >> 32);
        }
        BZ_ST(L.sh.done, 1);
#undef BZ_FAIL
#undef BZ_EOFCHK
    } else {
        // ================= consumer =================
        uint32_t n = 0;
        int err = SEMZ_OK;
        // wait for the MTF alphabet (or for a header that failed)
        while (!BZ_LD(L.sh.hdr_ready) && !BZ_LD(L.sh.done)) __builtin_amdgcn_s_sleep(8);
        if (BZ_LD(L.sh.hdr_ready)) {
            // inverse-MTF list: lane i of mtf_a holds entry i < 64, mtf_b .. mtf_d entries 64 .. 255
            uint32_t mtf_a = L.symbols[lane], mtf_b = L.symbols[64 + lane], mtf_c = L.symbols[128 + lane], mtf_d = L.symbols[192 + lane];
            uint32_t ab_len = 0, ab_val = 0, consumed = 0, produced_seen = 0;
            // mtfq entry: MTF index (8 bits) | length of the run in front of it << 8 (at most 2^20 - 1), or:
            constexpr uint32_t CV_LONG = 1u << 31;       // more run digits than +bzip2-ab-bound+
            constexpr uint32_t RUN_INLINE = 24;          // longer runs are stored by the whole wave, one run at a time
            // stores `run` copies of `front` at pre[at ..), all lanes
            auto emit_run = [&](uint32_t at, uint32_t front, uint32_t run) {
                BZ_COUNT(4);
                // the counter goes through readfirstlane so that the loop keeps a uniform exit: behind a divergent
                // exit every scalar that is live across the loop would count as divergent
                for (uint32_t k0 = 0; k0 < run; k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k0 + 64)))
                    if (k0 + lane < run) pre[at + k0 + lane] = (uint8_t)front;
                __builtin_amdgcn_wave_barrier();
            };
            // lane 0 takes `c0`, lane l > 0 takes lane l - 1 (the DPP shift leaves `old` in lane 0)
            auto shr1 = [](uint32_t v, uint32_t c0) {
                return (uint32_t)__builtin_amdgcn_update_dpp((int)c0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            };
            BZ_STAMP(4);
            for (;;) {
                uint32_t avail = produced_seen - consumed;
                if (avail < 64) {
                    for (;;) {
                        const uint32_t dn = BZ_LD(L.sh.done);            // read before `produced`: final if set
                        produced_seen = BZ_LD(L.sh.produced);
                        avail = produced_seen - consumed;
                        if (avail >= 64 || dn) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    BZ_STAMP(4);
                }
                const uint32_t m = avail < 64 ? avail : 64;
                if (m == 0) break;                                       // done and drained
                const uint32_t sq = (uint32_t)lane < m ? L.symq[(consumed + lane) & (BZ_QCAP - 1)] : 0u;
                consumed += m;
                if ((consumed & 2047) < 64) semz_take_turn(15);          // (semz_device.h; every 2048 symbols)
                const uint64_t validm = m == 64 ? ~0ull : (1ull << m) - 1;
                const uint64_t bmask = __builtin_amdgcn_ballot_w64(sq == 1) & validm;
                const uint64_t mtfmask = __builtin_amdgcn_ballot_w64(sq >= 2) & validm;
                BZ_ST(L.sh.consumed, consumed);                          // sq is in registers: the slots are free
                const uint32_t cnt = (uint32_t)__builtin_popcountll(mtfmask);
                // Every lane with an MTF symbol works out the run that ends right below it inside this batch: the
                // run symbols between the MTF symbol before it (position lastpos - 1) and itself are the digits of
                // the run length, RUNB digits straight out of the ballot (bzip2.lisp:221-243).
                const uint64_t lt = (1ull << lane) - 1;
                const uint64_t below = mtfmask & lt;
                const uint32_t lastpos = below ? 64u - (uint32_t)__builtin_clzll(below) : 0u;
                const uint32_t rl = (uint32_t)lane - lastpos;
                const uint32_t digits = (uint32_t)(bmask >> lastpos) & ((1u << (rl & 31)) - 1);
                const uint32_t runv = rl == 0 ? 0u : rl > (uint32_t)BZ_AB_BOUND ? CV_LONG : ((digits | (1u << rl)) - 1) << 8;
                // the MTF symbols squeezed together: lane e gets the e-th of them
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mtfmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mtfmask, 0));
                lds_sync();
                if (__builtin_amdgcn_inverse_ballot_w64(mtfmask)) L.mtfq[rank] = (sq - 1) | runv;
                __builtin_amdgcn_wave_barrier();
                lds_sync();
                const uint32_t cv = (uint32_t)lane < cnt ? L.mtfq[lane] : 0u;
                uint32_t run = (cv >> 8) & 0x1FFFFFu;
                bool toolong = (cv & CV_LONG) != 0;
                // the first MTF symbol also ends the run that was pending from the batches before
                if (cnt && ab_len) {
                    const uint32_t i0 = (uint32_t)__builtin_ctzll(mtfmask);
                    ab_val |= ((uint32_t)bmask & ((1u << i0) - 1)) << (ab_len & 31);
                    ab_len += i0;
                    const bool long0 = ab_len > (uint32_t)BZ_AB_BOUND;
                    const uint32_t run0 = long0 ? 0u : (ab_val | (1u << (ab_len & 31))) - 1;
                    ab_len = 0;
                    ab_val = 0;
                    run = lane == 0 ? run0 : run;
                    toolong = lane == 0 ? long0 : toolong;
                }
                // ---- the MTF walk proper: nothing but the list.  A run repeats the byte in front of the list, which
                // is the byte of the MTF symbol before it, so runs need no place in the walk: they are laid out
                // afterwards by all lanes.  x of the e-th symbol collects in lane e of xv.
                const uint32_t front0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mtf_a);
                const uint32_t idxv = cv & 0xFF;
                uint32_t xv = 0;
                if (__builtin_amdgcn_ballot_w64(idxv >= 64) == 0) {
                    // every index of the batch lies inside the first register: a loop without any other path, the
                    // wave's chain, 900 k times per block, and one block of assembler: eight instructions per symbol.
                    // m0 runs one ahead of the symbol (the index of the next symbol is read with it, five
                    // instructions before it is used as a lane select), so the byte of symbol e lands in lane e + 1
                    // of xv (symbol 63: lane 0) and one rotation puts the batch right afterwards.
#define BZ_MTF_STEP                                                                         \
    "s_add_u32 m0, m0, 1\n\t"                                                               \
    "v_readlane_b32 %[x], %[a], %[i]\n\t"                                                    \
    "v_cmp_ge_u32_e32 vcc, %[i], %[lane]\n\t"                                                \
    "v_readlane_b32 %[i], %[iv], m0\n\t"                                                     \
    "v_mov_b32_dpp %[t], %[a] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                     \
    "v_writelane_b32 %[xv], %[x], m0\n\t"                                                    \
    "v_writelane_b32 %[t], %[x], 0\n\t"                                                      \
    "v_cndmask_b32_e32 %[a], %[a], %[t], vcc\n\t"
                    uint32_t x, i, t, lim;
                    asm volatile("s_mov_b32 m0, 0\n\t"
                                 "v_readlane_b32 %[i], %[iv], 0\n\t"
                                 "s_sub_u32 %[lim], %[cnt], 3\n\t"
                                 "s_cmp_lt_u32 %[cnt], 4\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_nop 0\n"
                                 "1:\n\t"
                                 BZ_MTF_STEP BZ_MTF_STEP BZ_MTF_STEP BZ_MTF_STEP
                                 "s_cmp_lt_u32 m0, %[lim]\n\t"
                                 "s_cbranch_scc1 1b\n"
                                 "2:\n\t"
                                 "s_cmp_lt_u32 m0, %[cnt]\n\t"
                                 "s_cbranch_scc0 4f\n"
                                 "3:\n\t"
                                 BZ_MTF_STEP
                                 "s_cmp_lt_u32 m0, %[cnt]\n\t"
                                 "s_cbranch_scc1 3b\n"
                                 "4:"
                                 : [a] "+v"(mtf_a), [xv] "+v"(xv), [x] "=&s"(x), [i] "=&s"(i), [t] "=&v"(t), [lim] "=&s"(lim)
                                 : [iv] "v"(idxv), [lane] "v"(lane), [cnt] "s"(cnt)
                                 : "vcc", "scc", "m0");
#undef BZ_MTF_STEP
                    xv = (uint32_t)__builtin_amdgcn_update_dpp((int)xv, (int)xv, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
                } else {
                    // Indexes of 64 and more (incompressible data: three symbols in four): the whole list is in four
                    // registers, entry 64 k + l in lane l of register k.  The entries in front of the index move up by
                    // one: whole registers shift with the last lane of the register before them coming in at lane 0,
                    // the register that holds the index shifts up to the index's lane.  (With entries 64.. in LDS
                    // a symbol cost three LDS round trips: 192 ms per launch for blocks of random bytes, now 99.)
                    // One block of assembler with the four cases inside (written as four blocks under C++ branches the
                    // compiler copied the five registers around every one of them): the carries are taken first, the
                    // registers shift in place (lane 0 keeps its value), v_writelane puts what comes in at lane 0,
                    // and the symbol's byte goes into lane e of xv.  (Wait states: a DPP source must not have been
                    // written by the two instructions before it; no lane select here comes out of a vector
                    // instruction less than five instructions earlier.)
#define BZ_DPP_SHR(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define BZ_DPP_TMP(r) "v_mov_b32_dpp %[t], " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                    for (uint32_t e = 0; e < cnt; e++) {
                        const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)idxv, (int)e);
                        uint32_t x, c0, c1, c2, j, t;
                        asm volatile("s_mov_b32 m0, %[e]\n\t"
                                     "s_and_b32 %[j], %[idx], 63\n\t"
                                     "v_cmp_ge_u32_e32 vcc, %[j], %[lane]\n\t"
                                     "s_cmp_lt_u32 %[idx], 128\n\t"
                                     "s_cbranch_scc0 2f\n\t"
                                     "s_cmp_lt_u32 %[idx], 64\n\t"
                                     "s_cbranch_scc0 1f\n\t"
                                     // 0 .. 63
                                     "v_readlane_b32 %[x], %[a], %[j]\n\t"
                                     BZ_DPP_TMP("%[a]")
                                     "v_writelane_b32 %[t], %[x], 0\n\t"
                                     "v_writelane_b32 %[xv], %[x], m0\n\t"
                                     "v_cndmask_b32_e32 %[a], %[a], %[t], vcc\n\t"
                                     "s_branch 9f\n"
                                     "1:\n\t"       // 64 .. 127
                                     "v_readlane_b32 %[x], %[b], %[j]\n\t"
                                     "v_readlane_b32 %[c0], %[a], 63\n\t"
                                     BZ_DPP_TMP("%[b]") BZ_DPP_SHR("%[a]")
                                     "v_writelane_b32 %[t], %[c0], 0\n\t"
                                     "v_writelane_b32 %[xv], %[x], m0\n\t"
                                     "v_writelane_b32 %[a], %[x], 0\n\t"
                                     "v_cndmask_b32_e32 %[b], %[b], %[t], vcc\n\t"
                                     "s_branch 9f\n"
                                     "2:\n\t"
                                     "s_cmp_lt_u32 %[idx], 192\n\t"
                                     "s_cbranch_scc0 3f\n\t"
                                     // 128 .. 191
                                     "v_readlane_b32 %[x], %[c], %[j]\n\t"
                                     "v_readlane_b32 %[c0], %[a], 63\n\t"
                                     "v_readlane_b32 %[c1], %[b], 63\n\t"
                                     BZ_DPP_TMP("%[c]") BZ_DPP_SHR("%[a]") BZ_DPP_SHR("%[b]")
                                     "v_writelane_b32 %[t], %[c1], 0\n\t"
                                     "v_writelane_b32 %[xv], %[x], m0\n\t"
                                     "v_writelane_b32 %[a], %[x], 0\n\t"
                                     "v_writelane_b32 %[b], %[c0], 0\n\t"
                                     "v_cndmask_b32_e32 %[c], %[c], %[t], vcc\n\t"
                                     "s_branch 9f\n"
                                     "3:\n\t"       // 192 .. 255
                                     "v_readlane_b32 %[x], %[d], %[j]\n\t"
                                     "v_readlane_b32 %[c0], %[a], 63\n\t"
                                     "v_readlane_b32 %[c1], %[b], 63\n\t"
                                     "v_readlane_b32 %[c2], %[c], 63\n\t"
                                     BZ_DPP_TMP("%[d]") BZ_DPP_SHR("%[a]") BZ_DPP_SHR("%[b]") BZ_DPP_SHR("%[c]")
                                     "v_writelane_b32 %[t], %[c2], 0\n\t"
                                     "v_writelane_b32 %[xv], %[x], m0\n\t"
                                     "v_writelane_b32 %[a], %[x], 0\n\t"
                                     "v_writelane_b32 %[b], %[c0], 0\n\t"
                                     "v_writelane_b32 %[c], %[c1], 0\n\t"
                                     "v_cndmask_b32_e32 %[d], %[d], %[t], vcc\n"
                                     "9:"
                                     : [a] "+v"(mtf_a), [b] "+v"(mtf_b), [c] "+v"(mtf_c), [d] "+v"(mtf_d), [xv] "+v"(xv), [x] "=&s"(x),
                                       [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [j] "=&s"(j), [t] "=&v"(t)
                                     : [idx] "s"(idx), [e] "s"(e), [lane] "v"(lane) : "vcc", "scc", "m0");
                    }
#undef BZ_DPP_SHR
#undef BZ_DPP_TMP
                }
                // ---- layout: symbol e puts its run, then its byte
                const bool live = (uint32_t)lane < cnt;
                const uint32_t incl = wave_scan_add(live ? 1u + run : 0u);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t xpos = n + incl - 1;                      // where the symbol's own byte goes
                // failures, first in stream order: run digits, run size, then a symbol that arrives with the block full
                const uint64_t longm = __builtin_amdgcn_ballot_w64(live && toolong);
                if (longm || n + total > block_size) {
                    const uint64_t sizem = __builtin_amdgcn_ballot_w64(live && run && xpos > block_size);
                    const uint64_t fullm = __builtin_amdgcn_ballot_w64(live && xpos >= block_size);
                    const uint64_t anym = longm | sizem | fullm;
                    if (anym) {
                        const uint64_t first = anym & (0 - anym);
                        err = (longm & first) ? SEMZ_E_BZ_AB_RUN_LENGTH : (sizem & first) ? SEMZ_E_BZ_AB_RUN_SIZE : SEMZ_E_BZ_TOO_MANY_CODES;
                        break;
                    }
                }
                // ---- stores
                const uint32_t prevx = shr1(xv, front0);
                if (live) pre[xpos] = (uint8_t)xv;
                const bool inl = live && run <= RUN_INLINE;
                for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(inl && k < run); k = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k + 1)))
                    if (inl && k < run) pre[xpos - run + k] = (uint8_t)prevx;
                __builtin_amdgcn_wave_barrier();
                uint64_t bigm = __builtin_amdgcn_ballot_w64(live && run > RUN_INLINE);
                while (bigm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(bigm);
                    bigm &= bigm - 1;
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)run, (int)j);
                    emit_run((uint32_t)__builtin_amdgcn_readlane((int)xpos, (int)j) - r, (uint32_t)__builtin_amdgcn_readlane((int)prevx, (int)j), r);
                }
                n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n + total));
                {
                    // run symbols at the end of the batch stay pending
                    const uint32_t lp = mtfmask ? 64 - (uint32_t)__builtin_clzll(mtfmask) : 0;
                    const uint32_t rl2 = m - lp;
                    if (ab_len + rl2 > (uint32_t)BZ_AB_BOUND) err = SEMZ_E_BZ_AB_RUN_LENGTH;
                    else if (rl2) {
                        ab_val |= ((uint32_t)(bmask >> lp) & ((1u << rl2) - 1)) << ab_len;
                        ab_len += rl2;
                    }
                }
                BZ_STAMP(2);
                if (err != SEMZ_OK) break;
            }
            if (err != SEMZ_OK) BZ_ST(L.sh.abort, 1);
            while (!BZ_LD(L.sh.done)) __builtin_amdgcn_s_sleep(8);
            if (err == SEMZ_OK && (int)L.sh.status == SEMZ_OK && ab_len) {
                // the run in front of the end-of-block symbol
                const uint32_t run = (ab_val | (1u << ab_len)) - 1;
                if (n + run > block_size) err = SEMZ_E_BZ_AB_RUN_SIZE;
                else { emit_run(n, (uint32_t)__builtin_amdgcn_readfirstlane((int)mtf_a), run); n += run; }
            }
        }
        // `done` has been seen: the producer's fields are final
        int status = err != SEMZ_OK ? err : (int)BZ_LD(L.sh.status);
        uint32_t detail = err != SEMZ_OK ? 0u : BZ_LD(L.sh.detail);
        const uint32_t orig = BZ_LD(L.sh.orig);
        // decode-bzip2-bwt's precondition; deliberate divergence, see DESIGN.md
        if (status == SEMZ_OK && (n == 0 ? orig >= block_size : orig >= n)) { status = SEMZ_E_BZ_ORIGPTR; detail = orig; }
        if (lane == 0) {
            bz_meta m;
            m.n = n; m.orig = orig; m.randomized = L.sh.randomized; m.want_crc = L.sh.want_crc;
            m.status = status; m.detail = detail;
            m.end_bit = ((uint64_t)L.sh.end_hi << 32) | L.sh.end_lo;
            m.out_len = 0; m.out_off = 0;
            meta[u] = m;
        }
    }
#ifdef SEMZ_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&g_bz_stamps[i], (unsigned long long)t_acc[i]);
        for (int i = 0; i < 6; i++) atomicAdd(&g_bz_stamps[5 + i], (unsigned long long)n_acc[i]);
        if (wid == 1) atomicAdd(&g_bz_stamps[11], 1ull);
    }
#endif
#undef BZ_LD
#undef BZ_ST
}

// ---- RLE1 state machine (decode-bzip2-rle1, bzip2.lisp:59-102) ----
struct Rle1 { int last; int reps; };
// consumes one post-BWT byte; returns how many output bytes it produces and which byte
__device__ __forceinline__ uint32_t rle1_step(Rle1 &s, uint32_t b, uint32_t &outbyte)
{
    if (s.reps < 3) {
        s.reps = ((int)b == s.last) ? s.reps + 1 : 0;
        s.last = (int)b;
        outbyte = b;
        return 1;
    }
    outbyte = (uint32_t)s.last;
    s.last = -1;
    s.reps = 0;
    return b;
}

// Marker spacing of a block of n bytes: the smallest power of two with S * BZ_CHAINS >= n.
__device__ __forceinline__ uint32_t bz_spacing(uint32_t n)
{
    uint32_t S = 1;
    while ((uint64_t)S * BZ_CHAINS < n) S <<= 1;
    return S;
}

__global__ __launch_bounds__(BWT_THREADS) void bz_sort_kernel(uint32_t nunits, const bz_meta *meta, const uint8_t *pre_all,
                                                               uint32_t *tt_all)
{
    __shared__ __align__(16) uint32_t smem[BWT_WAVES * 256 + 256];
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const bz_meta m = meta[u];
    if (m.status != SEMZ_OK) return;
    const uint32_t n = m.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *pre = pre_all + (size_t)u * BZ_STRIDE;
    uint32_t *tt = tt_all + (size_t)u * BZ_STRIDE;
    // ---- 1. per-wave byte histograms ----
    uint32_t *hist = smem;                         // [16][256]
    uint32_t *cumm = smem + BWT_WAVES * 256;       // [256]
    for (uint32_t i = tid; i < BWT_WAVES * 256 + 256; i += BWT_THREADS) smem[i] = 0;
    __syncthreads();
    const uint32_t chunk = (((n + BWT_WAVES - 1) / BWT_WAVES) + 63) & ~63u;
    const uint32_t cbeg = wave * chunk, cend = cbeg + chunk < n ? cbeg + chunk : n;
    // (eight 64-byte pieces are loaded before any is used: one wave reading its chunk piece by piece is bound by
    // the latency of each load otherwise)
    for (uint32_t b0 = cbeg; b0 < cend; b0 += 512) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = b0 + 64 * q + lane;
            v[q] = i < cend ? pre[i] : 0x100u;
        }
        // (this is the last column of a BWT: the 64 bytes of a piece are more often than not one and the same, and
        // 64 atomics on one counter take their turns)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[q]);
            if (__ballot(v[q] != v0) == 0) {
                if (lane == 0 && v0 < 0x100u) atomicAdd(&hist[wave * 256 + v0], 64u);
            } else if (v[q] < 0x100u) atomicAdd(&hist[wave * 256 + v[q]], 1u);
        }
    }
    __syncthreads();
    // ---- 2. prefix sums: cumm[v] (bzip2.lisp:150-151) and each wave's base per byte value ----
    if (tid < 256) {
        uint32_t total = 0;
        for (int w = 0; w < BWT_WAVES; w++) { uint32_t t = hist[w * 256 + tid]; hist[w * 256 + tid] = total; total += t; }
        cumm[tid] = total;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int v = 0; v < 256; v++) { uint32_t t = cumm[v]; cumm[v] = acc; acc += t; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < BWT_WAVES * 256; i += BWT_THREADS) hist[i] += cumm[i & 255];
    __syncthreads();
    // ---- 3. stable scatter: tt[cumm[v]++] = (i << 8) | v  (bzip2.lisp:153-156) ----
    for (uint32_t b8 = cbeg; b8 < cend; b8 += 512) {
        uint32_t vq[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = b8 + 64 * q + lane;
            vq[q] = i < cend ? pre[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t b0 = b8 + 64 * q;
            if (b0 >= cend) break;
            const uint32_t i = b0 + lane;
            const bool valid = i < cend;
            const uint32_t v = vq[q];
            uint64_t same = __ballot(valid);
            if (__ballot(!valid || v != (uint32_t)__builtin_amdgcn_readfirstlane((int)v)) != 0) {    // (not all the same byte)
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    uint64_t bm = __ballot((v >> bit) & 1);
                    same &= ((v >> bit) & 1) ? bm : ~bm;
                }
            }
            const uint32_t rank = __popcll(same & ((1ull << lane) - 1));
            const uint32_t cnt = __popcll(same);
            uint32_t old = 0;
            if (valid) old = hist[wave * 256 + v];
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                tt[old + rank] = (i << 8) | v;
                if (rank == 0) hist[wave * 256 + v] = old + cnt;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    }
}

// ---- 4. inverse BWT walk (bzip2.lisp:158-163) ----
// out[j] = low byte of tt[k_j], k_0 = origPtr, k_{j+1} = tt[k_j] >> 8: one dependent random 4-byte read per output
// byte.  Markers: k = origPtr (id 0) and k = j S (id j + 1); every marker walks until it meets a marker, its bytes go
// to row `id` of the block's `temp` (4 S bytes; chain lengths are spread geometrically around S), and a chain that
// fills its row goes on in a fresh row under a fresh id.  rows[id] = {bytes, following row}.
//
// What bounds the walk is where the vector lives.  With every block of a batch walked at once (the first version)
// each read fetched a whole 128-byte line from HBM, 71 GB for 540 MB of output.  Here an XCD walks one block at a
// time, more or less: a wave takes chain starts from the counter of the block its XCD is at and moves to the XCD's
// next block (+ 8) when that counter runs out, a lane whose chain has ended takes the next start at once.  Chains
// are short (S = 64) so that the lanes still busy with a block when the others have moved on hold the smaller
// part of its steps, and the vector of the block in hand (3.6 MB) mostly stays in the XCD's 4 MB L2: 83 % of the
// kernel's L2 requests hit (TCC_HIT / TCC_REQ), where 94 % missed before.  Measured on the 600-block bench:
// S = 16 / 64 / 256 walk in 8.1 / 8.6 / 12.6 ms and link in 8.0 / 5.2 / 6.0 ms (short rows waste lines there).
__global__ __launch_bounds__(WALK_THREADS) void bz_walk_kernel(uint32_t nunits, const bz_meta *meta, const uint32_t *tt_all,
                                                                uint8_t *temp_all, bz_row *rows_all, uint32_t *starts, uint32_t *extra)
{
    // which XCD this workgroup landed on (XCC_ID, hardware register 20, bits 3:0).  Only the affinity hangs on it:
    // a wave that finds no block left on its XCD's list goes through the other lists, so every block is walked
    // wherever the workgroups were placed.
    const uint32_t home = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) % BZ_XCDS;
    const uint32_t lane = threadIdx.x & 63;
    // A chain's bytes collect in a row of LDS (256 bytes per lane; word w of thread t's row sits at (w + t) mod 64,
    // so neither the lanes' own stores nor the wave's read of one row collide) and go out in one piece when the
    // chain ends: with every lane storing a dword per four steps into a row of its own in HBM, the lines of the
    // 8192 rows in progress (2 MB) lived in the L2 beside the vector and two thirds of the vector's reads missed.
    __shared__ uint32_t rowbuf[WALK_THREADS * 64];
    const uint32_t tid = threadIdx.x, wbase = threadIdx.x & ~63u;
    uint32_t *myrow = rowbuf + tid * 64;
    // starts[b]: chain starts of block b handed out; extra[b]: ids handed out beyond its markers
    // the block in hand (wave-uniform)
    uint32_t cur = home, lists = 0;                // block index; lists (XCDs) finished so far
    bool gated = false;                            // the block in hand has no starts left: see below
    uint32_t c_nmark = 0, c_S = 1, c_orig = 0;
    uint32_t next = 0, end = 0;                    // this wave's share of the block's starts
    bool exhausted = false, loaded = false;
    // the chain in hand (per lane)
    bool active = false, pend = false, cont = false;   // walking; row complete and not yet out; the chain goes on behind it
    uint32_t k = 0, len = 0, id = 0, acc = 0, smask = 0, sshift = 0, orig = 0, cap = 0, nmark = 0, blk = 0, nxt = 0;
    const uint32_t *tt = tt_all;
    bz_row *rows = nullptr;
    for (;;) {
        // complete rows out, the whole wave per row: one or two full lines at a time
        uint64_t pm = __ballot(pend);
        if (pm) {
            lds_sync();
            do {
                const uint32_t j = (uint32_t)__builtin_ctzll(pm);
                pm &= pm - 1;
                const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)id, (int)j), lenj = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)j);
                const uint32_t blkj = (uint32_t)__builtin_amdgcn_readlane((int)blk, (int)j), capj = (uint32_t)__builtin_amdgcn_readlane((int)cap, (int)j);
                uint32_t *dst = reinterpret_cast<uint32_t *>(temp_all + (size_t)blkj * BZ_TEMP_STRIDE + (size_t)idj * capj);
                const uint32_t w = rowbuf[(wbase + j) * 64 + ((lane + wbase + j) & 63)];
                if (4 * lane < lenj) __builtin_nontemporal_store(w, dst + lane);
            } while (pm);
            lds_sync();
            if (pend) {
                pend = false;
                if (!cont) put_row(&rows[id], len, nxt);
                else {
                    cont = false;
                    const uint32_t nid = nmark + atomicAdd(&extra[blk], 1u);
                    if (nid < (uint32_t)BZ_IDS) {
                        put_row(&rows[id], cap, nid);
                        id = nid;
                        len = 0;
                        acc = 0;
                        active = true;
                    } else put_row(&rows[id], cap, 0xFFFFFFFFu);   // cannot happen (BZ_IDS covers n / S + 2 + n / (4 S) + 2); ends the list
                }
            }
        }
        const uint64_t idle = __ballot(!active);
        if (idle) {
            while (next >= end && !exhausted) {
                // Two vectors do not fit one L2: with the lanes that run out of starts going straight on to the
                // next block, both blocks' reads missed for the length of the old block's last chains (seven times
                // the lines a block has were fetched: TCC_EA0_RDREQ, one block per XCD against 75).  So a wave
                // waits at the door of the next block until few of its own lanes still walk; the waves of an XCD
                // are alike, so the old block is then nearly done everywhere (a count of finished chains in memory
                // says it exactly and costs three times what it saves: atomics on one address).  With 8192 chains
                // of 64 steps in flight the end of a block is more than half of it, so waiting costs what the
                // misses cost: fabric reads / walk time on the 600-block bench are 238 M / 7.0 ms without the door,
                // 193 M / 7.05 with 32 lanes, 166 M / 7.3 with 24, 130 M / 7.7 with 16, 89 M / 8.4 with 8 (and
                // 46 M / 8.2 with markers every 32 positions, where bz_link_kernel pays 0.7 ms for the rows).
                if (gated) {
                    if (__popcll(__ballot(active)) > WALK_KEEP) break;
                    gated = false;
                }
                if (cur >= nunits) {               // this list is done: on to the next XCD's
                    if (++lists == (uint32_t)BZ_XCDS) { exhausted = true; break; }
                    cur = (home + lists) % BZ_XCDS;
                    loaded = false;
                    continue;
                }
                if (!loaded) {
                    const uint32_t n = meta[cur].n;
                    c_S = bz_spacing(n);
                    c_nmark = meta[cur].status == SEMZ_OK && n ? (n + c_S - 1) / c_S + 1 : 0;
                    c_orig = meta[cur].orig;
                    loaded = true;
                }
                uint32_t base = 0xFFFFFFFFu;
                if (c_nmark) {
                    if (lane == 0) base = atomicAdd(&starts[cur], (uint32_t)WALK_GRAB);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                }
                if (base >= c_nmark) {
                    gated = c_nmark != 0;
                    cur += BZ_XCDS;
                    loaded = false;
                } else {
                    next = base;
                    end = base + WALK_GRAB < c_nmark ? base + WALK_GRAB : c_nmark;
                }
            }
            const uint32_t room = end - next;
            if (!active) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0));
                if (rank < room) {
                    const uint32_t c = next + rank;
                    // (when origPtr is a marker position the marker's own chain would be a twin of chain 0 that nothing
                    // links to; it is not walked: its rows, and above all the rows it would go on in under ids handed
                    // out at run time, must not exist for bz_link_kernel to come across)
                    active = c == 0 || (c - 1) * c_S != c_orig;
                    blk = cur;
                    nmark = c_nmark;
                    smask = c_S - 1;
                    sshift = (uint32_t)__builtin_ctz(c_S);
                    cap = 4 * c_S;
                    orig = c_orig;
                    id = c;
                    k = c == 0 ? c_orig : (c - 1) * c_S;
                    len = 0;
                    acc = 0;
                    tt = tt_all + (size_t)cur * BZ_STRIDE;
                    rows = rows_all + (size_t)cur * BZ_IDS;
                }
            }
            const uint32_t want = (uint32_t)__popcll(idle);
            next += want < room ? want : room;
            if (exhausted && __ballot(active) == 0) break;
        }
        // a few steps between two looks at the idle lanes (a chain that ends waits two steps on average)
#pragma unroll
        for (int st = 0; st < WALK_STEPS; st++) {
            if (active) {
                const uint32_t v = tt[k];
                acc |= (v & 0xFF) << (8 * (len & 3));
                if ((len & 3) == 3) { myrow[((len >> 2) + tid) & 63] = acc; acc = 0; }
                k = v >> 8;
                len++;
                const bool fin = k == orig || (k & smask) == 0;
                if (fin || len == cap) {
                    if (len & 3) myrow[((len >> 2) + tid) & 63] = acc;
                    nxt = k == orig ? 0u : (k >> sshift) + 1;
                    cont = !fin;
                    pend = true;
                    active = false;
                }
            }
        }
    }
}

// The rows in walk order, to their places in `post`.  There are tens of thousands of rows per block, so the list
// is ranked in two levels: every id that is a multiple of BZ_SUPER heads a run of rows (id 0, origPtr's row, is
// one of them); a thread follows each run to the next head and sums its bytes, one thread puts the runs in
// order, and then the rows are copied side by side, a wave per row: on its way along a run the thread has noted for
// every row which run it belongs to and where in the run it starts (`info`, two words per row), so the copy has no list
// to follow (a thread copying the rows of its runs one dword after the other took 3 of the kernel's 4 ms).
__global__ __launch_bounds__(BWT_THREADS) void bz_link_kernel(uint32_t nunits, bz_meta *meta, uint8_t *post_all,
                                                               bz_seg *seg_all, const uint8_t *temp_all, const bz_row *rows_all,
                                                               const uint32_t *extra, uint2 *info_all)
{
    constexpr uint32_t LINK_WORDS = 3 * (BZ_IDS / BZ_SUPER + 1) + 16 > 10 * BZ_SEGS ? 3 * (BZ_IDS / BZ_SUPER + 1) + 16 : 10 * BZ_SEGS;
    __shared__ __align__(16) uint32_t smem[LINK_WORDS];
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const bz_meta m = meta[u];
    if (m.status != SEMZ_OK) return;
    const uint32_t n = m.n, tid = threadIdx.x;
    uint8_t *post = post_all + (size_t)u * BZ_STRIDE;
    bz_seg *segs = seg_all + (size_t)u * BZ_SEGS;
    const uint8_t *temp = temp_all + (size_t)u * BZ_TEMP_STRIDE;
    const bz_row *rows = rows_all + (size_t)u * BZ_IDS;
    constexpr uint32_t NRUN = BZ_IDS / BZ_SUPER + 1;
    uint32_t *run_len = smem;                      // [NRUN] bytes in the run headed by id BZ_SUPER * r
    uint32_t *run_next = smem + NRUN;              // the run that follows (0xFFFFFFFF: the list breaks off)
    uint32_t *run_off = smem + 2 * NRUN;           // where the run's bytes go; 0xFFFFFFFF = not on origPtr's cycle
    uint32_t *misc = smem + 3 * NRUN;              // [0] = cycle length
    const uint32_t S = bz_spacing(n);
    const uint32_t CAP = 4 * S;
    uint32_t nids = n ? (n + S - 1) / S + 1 + extra[u] : 0;
    if (nids > (uint32_t)BZ_IDS) nids = BZ_IDS;
    const uint32_t nrun = (nids + BZ_SUPER - 1) / BZ_SUPER;
    uint2 *info = info_all + (size_t)u * BZ_IDS;
    for (uint32_t i = tid; i < nids; i += BWT_THREADS) info[i] = make_uint2(0xFFFFFFFFu, 0);   // (rows no run reaches are on no cycle with a head)
    __threadfence_block();
    __syncthreads();
    // When origPtr is a marker position two chains start there: id 0 and the marker's own.  Nothing links to the
    // second one (a chain that meets origPtr is followed by id 0), but it links into the list like its twin, so a
    // run that begins with it would claim the rows behind it a second time: it is left out.
    const uint32_t twin = (m.orig & (S - 1)) == 0 ? m.orig / S + 1 : 0xFFFFFFFFu;
    for (uint32_t r = tid; r < nrun; r += BWT_THREADS) {
        uint32_t id = r * BZ_SUPER, sum = 0, nx = 0xFFFFFFFFu;
        for (uint32_t guard = 0; guard < nids && r * BZ_SUPER != twin; guard++) {   // (a run ends at the next head; the guard is belt and braces)
            const bz_row w = rows[id];
            info[id] = make_uint2(r, sum);                         // (a run is sixteen rows on average, not at most)
            sum += w.len;
            nx = w.next;
            if (nx >= nids) { nx = 0xFFFFFFFFu; break; }
            if (nx % BZ_SUPER == 0) { nx /= BZ_SUPER; break; }
            id = nx;
            nx = 0xFFFFFFFFu;
        }
        run_len[r] = sum;
        run_next[r] = nx;
        run_off[r] = 0xFFFFFFFFu;
    }
    __threadfence_block();                         // (`info` is in global memory: __syncthreads() by itself only orders LDS)
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0, off = 0;
        while (off < n && t < nrun && run_off[t] == 0xFFFFFFFFu) {
            run_off[t] = off;
            off += run_len[t];
            t = run_next[t];
        }
        misc[0] = off;                              // >= n, or the length of origPtr's cycle if shorter
    }
    __syncthreads();
    const uint32_t cyc = misc[0];
    const uint32_t wlim = cyc < n ? cyc : n;
    // The rows, 64 per wave and trip: every lane looks up where its row goes, then the wave copies the rows one
    // after the other, a dword per lane to whatever address it belongs at (the hardware takes unaligned dword
    // stores), single bytes at a row's end.
    {
        const uint32_t lane = tid & 63;
        for (uint32_t base = tid & ~63u; base < nids; base += BWT_THREADS) {
            const uint32_t id = base + lane;
            uint32_t ln = 0, so = 0;
            if (id < nids) {
                const uint2 inf = info[id];
                const uint32_t ro = inf.x != 0xFFFFFFFFu ? run_off[inf.x] : 0xFFFFFFFFu;
                if (ro != 0xFFFFFFFFu && inf.y < wlim && ro < wlim - inf.y) {
                    so = ro + inf.y;
                    const uint32_t l = rows[id].len;
                    ln = l < CAP ? l : CAP;
                    if (so >= wlim) ln = 0;
                    else if (so + ln > wlim) ln = wlim - so;
                }
            }
            // (LINK_DEPTH rows in flight: with a row's store behind its load, one after the other, the copy was one
            // memory latency per row, 1.9 ms of the kernel)
            uint64_t todo = __ballot(ln != 0);
            while (todo) {
                uint32_t x[LINK_DEPTH], lnv[LINK_DEPTH], sov[LINK_DEPTH];
#pragma unroll
                for (int q = 0; q < LINK_DEPTH; q++) {
                    const bool have = todo != 0;
                    const uint32_t j = have ? (uint32_t)__builtin_ctzll(todo) : 0u;
                    todo &= todo - (have ? 1 : 0);
                    lnv[q] = have ? (uint32_t)__builtin_amdgcn_readlane((int)ln, (int)j) : 0u;
                    sov[q] = (uint32_t)__builtin_amdgcn_readlane((int)so, (int)j);
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(temp + (size_t)(base + j) * CAP);
                    x[q] = 4 * lane < lnv[q] ? src[lane] : 0u;
                }
#pragma unroll
                for (int q = 0; q < LINK_DEPTH; q++) {
                    uint8_t *dst = post + sov[q];
                    // (no loop for the one to three bytes at a row's end: behind a loop the compiler waits for every
                    // store before the next one)
                    const uint32_t have = lnv[q] > 4 * lane ? lnv[q] - 4 * lane : 0u;
                    if (have >= 4) __builtin_memcpy(dst + 4 * lane, &x[q], 4);
                    if (have >= 1 && have < 4) dst[4 * lane] = (uint8_t)x[q];
                    if (have >= 2 && have < 4) dst[4 * lane + 1] = (uint8_t)(x[q] >> 8);
                    if (have == 3) dst[4 * lane + 2] = (uint8_t)(x[q] >> 16);
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (cyc < n) {
        // origPtr's cycle is shorter than the block: the output is periodic
        if (cyc)
            for (uint32_t j = cyc + tid; j < n; j += BWT_THREADS) post[j] = post[j % cyc];
        __threadfence_block();
        __syncthreads();
    }
    // ---- 5. de-randomise (bzip2.lisp:112-121) ----
    if (m.randomized && tid == 0) {
        uint32_t di = 0;
        for (uint32_t ri = 0;; ri++) {
            uint32_t skip = BZ_RAND[ri & 511];
            di = ri == 0 ? skip - 2 : di + skip;
            if (di >= n) break;
            post[di] ^= 1;
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- 6. RLE1 sizing: every thread parses its segment from each of the five
    //         possible entry classes; thread 0 then chains the real states ----
    // classes: 0 fresh (last != first byte, reps < 3), 1 count byte expected (reps == 3),
    //          2..4 last == first byte with reps 0..2
    uint32_t *r_len = smem;                         // [BZ_SEGS][5]
    uint32_t *r_exit = smem + BZ_SEGS * 5;          // [BZ_SEGS][5]  (last+1) << 2 | reps
    const uint32_t seglen = (n + BZ_SEGS - 1) / BZ_SEGS;
    for (uint32_t sg = tid; sg < (uint32_t)BZ_SEGS; sg += BWT_THREADS) {
        const uint32_t sbeg = sg * seglen < n ? sg * seglen : n;
        const uint32_t send = sbeg + seglen < n ? sbeg + seglen : n;
        const int first = sbeg < send ? post[sbeg] : -1;
        Rle1 st[5] = {{-1, 0}, {first, 3}, {first, 0}, {first, 1}, {first, 2}};
        uint32_t ol[5] = {0, 0, 0, 0, 0};
        // the segment, one whole 64-byte line at a time (see bz_rle1_kernel)
        for (uint32_t line = sbeg & ~63u; line < send; line += 64) {
            uint32_t w[16];
            const uint4 *lp = reinterpret_cast<const uint4 *>(post + line);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 x = lp[q];
                w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
            }
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const uint32_t i = line + j;
                if (i >= sbeg && i < send) {
                    const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFF;
                    uint32_t ob;
#pragma unroll
                    for (int c = 0; c < 5; c++) ol[c] += rle1_step(st[c], b, ob);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 5; c++) {
            r_len[sg * 5 + c] = ol[c];
            r_exit[sg * 5 + c] = ((uint32_t)(st[c].last + 1) << 2) | (uint32_t)st[c].reps;
        }
    }
    __syncthreads();
    if (tid == 0) {
        Rle1 st = {-1, 0};
        uint64_t off = 0;
        for (uint32_t s = 0; s < BZ_SEGS; s++) {
            uint32_t b0 = s * seglen < n ? s * seglen : n;
            bz_seg g;
            g.off = (uint32_t)off;
            g.last = (int16_t)st.last;
            g.reps = (int16_t)st.reps;
            segs[s] = g;
            if (b0 >= n) continue;
            int first = post[b0];
            int cls = st.reps == 3 ? 1 : (st.last == first ? 2 + st.reps : 0);
            off += r_len[s * 5 + cls];
            uint32_t ex = r_exit[s * 5 + cls];
            // class 1 with a one-byte segment keeps no history; other exits are self-contained
            st.last = (int)(ex >> 2) - 1;
            st.reps = (int)(ex & 3);
        }
        meta[u].out_len = off;
    }
}

// Exclusive scan of the block sizes for SEMZ_OUT_AUTO units, capacity checks, results and tails.  One workgroup:
// a thread sums the sizes of its stretch of the blocks, the stretches' sums are scanned through LDS, and the thread
// goes over its stretch again with the offset it starts at (a serial loop took 3.7 ms for 4800 blocks).
constexpr int OFFS_THREADS = 1024;
__global__ __launch_bounds__(OFFS_THREADS) void bz_offsets_kernel(const semz_unit *units, uint32_t nunits, bz_meta *meta, uint64_t out_cap,
                                                                  semz_result *results, semz_tail *tails)
{
    __shared__ uint64_t sums[OFFS_THREADS];
    if (blockIdx.x != 0) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nunits + OFFS_THREADS - 1) / OFFS_THREADS;
    const uint32_t lo = tid * per < nunits ? tid * per : nunits, hi = lo + per < nunits ? lo + per : nunits;
    uint64_t mine = 0;
    for (uint32_t u = lo; u < hi; u++)
        if (meta[u].status == SEMZ_OK && units[u].out_off == SEMZ_OUT_AUTO) mine += meta[u].out_len;
    sums[tid] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < OFFS_THREADS; d <<= 1) {          // inclusive scan
        const uint64_t add = tid >= d ? sums[tid - d] : 0;
        __syncthreads();
        sums[tid] += add;
        __syncthreads();
    }
    uint64_t auto_off = sums[tid] - mine;
    for (uint32_t u = lo; u < hi; u++) {
        bz_meta m = meta[u];
        semz_result r{};
        semz_tail t{};
        r.status = m.status;
        r.detail = m.detail;
        r.in_end_bit = m.end_bit;
        if (m.status == SEMZ_OK) {
            uint64_t off, cap;
            if (units[u].out_off == SEMZ_OUT_AUTO) { off = auto_off; cap = out_cap > off ? out_cap - off : 0; auto_off += m.out_len; }
            else { off = units[u].out_off; cap = units[u].out_cap; }
            if (m.out_len > cap) { r.status = SEMZ_E_OUT_CAP; m.status = SEMZ_E_OUT_CAP; }
            m.out_off = off;
            r.out_off = off;
            r.out_len = m.status == SEMZ_OK ? m.out_len : 0;
            t.check_kind = SEMZ_TK_BZCRC;
            t.want[0] = m.want_crc;
            t.have_check = 1;
            meta[u] = m;
        }
        results[u] = r;
        tails[u] = t;
    }
}

__global__ __launch_bounds__(1024) void bz_rle1_kernel(uint32_t nunits, const bz_meta *meta, const uint8_t *post_all,
                                                       const bz_seg *seg_all, uint8_t *out)
{
    // One thread per segment (their entry states and output offsets come from bz_bwt_kernel).  The threads of
    // a wave walk segments that lie ~900 bytes apart, so byte-wise loads and stores would touch 64 different
    // cache lines per instruction, each of them again for every byte.  Instead a thread takes its input one
    // whole 64-byte line at a time into registers, and collects its output in a 64-bit register that is
    // stored with one aligned 8-byte store.
    const uint32_t u = blockIdx.x;
    if (u >= nunits) return;
    const bz_meta m = meta[u];
    if (m.status != SEMZ_OK) return;
    const uint32_t n = m.n, tid = threadIdx.x;
    const uint8_t *post = post_all + (size_t)u * BZ_STRIDE;              // 64-byte aligned (BZ_STRIDE is a multiple of 64)
    const bz_seg g = seg_all[(size_t)u * BZ_SEGS + tid];
    uint8_t *o = out + m.out_off + g.off;
    const uint32_t seglen = (n + BZ_SEGS - 1) / BZ_SEGS;
    const uint32_t sbeg = tid * seglen < n ? tid * seglen : n;
    const uint32_t send = sbeg + seglen < n ? sbeg + seglen : n;
    Rle1 st = {g.last, g.reps};
    uint64_t acc = 0;            // pending output bytes, little-endian
    uint32_t nacc = 0;
    bool aligned = ((uintptr_t)o & 7) == 0;
    auto emit = [&](uint32_t byte, uint32_t k) {
        while (k) {
            if (!aligned) {
                *o++ = (uint8_t)byte;
                k--;
                aligned = ((uintptr_t)o & 7) == 0;
            } else if (nacc == 0 && k >= 8) {
                *reinterpret_cast<uint64_t *>(o) = 0x0101010101010101ull * byte;
                o += 8;
                k -= 8;
            } else {
                acc |= (uint64_t)byte << (8 * nacc);
                k--;
                if (++nacc == 8) {
                    *reinterpret_cast<uint64_t *>(o) = acc;
                    o += 8;
                    acc = 0;
                    nacc = 0;
                }
            }
        }
    };
    for (uint32_t line = sbeg & ~63u; line < send; line += 64) {
        uint32_t w[16];
        const uint4 *lp = reinterpret_cast<const uint4 *>(post + line);      // the scratch row is BZ_STRIDE long: the last line exists
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = lp[q];
            w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const uint32_t i = line + j;
            if (i >= sbeg && i < send) {
                uint32_t ob;
                const uint32_t k = rle1_step(st, (w[j >> 2] >> (8 * (j & 3))) & 0xFF, ob);
                emit(ob, k);
            }
        }
    }
    for (uint32_t q = 0; q < nacc; q++) o[q] = (uint8_t)(acc >> (8 * q));
}

} // namespace

#ifdef SEMZ_STAMPS
extern "C" int semz_debug_bz_stamps(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bz_stamps), sizeof(g_bz_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bz_stamps), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

size_t semz_bzip2_scratch_per_block()
{
    return sizeof(bz_meta) + (size_t)BZ_STRIDE * 2 + BZ_STRIDE * sizeof(uint32_t) + BZ_TEMP_STRIDE + BZ_SEGS * sizeof(bz_seg) +
           BZ_IDS * (sizeof(bz_row) + sizeof(uint2)) + 2 * sizeof(uint32_t) + 16;
}

int semz_run_bzip2(semz_ctx *c, const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                   uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails)
{
    void *meta, *pre, *post, *tt, *segs;
    int rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_BZ_META, (size_t)nunits * sizeof(bz_meta), &meta))) return rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_BZ_DATA, (size_t)nunits * BZ_STRIDE * 2, &pre))) return rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_BZ_TT, (size_t)nunits * (BZ_STRIDE * sizeof(uint32_t) + BZ_TEMP_STRIDE), &tt))) return rc;
    // RLE1 segment states, then the walk's rows (length, successor), then its counters (two per block)
    const size_t segs_bytes = (size_t)nunits * BZ_SEGS * sizeof(bz_seg);
    const size_t rows_bytes = (size_t)nunits * BZ_IDS * sizeof(bz_row);
    const size_t ctl_bytes = 2 * (size_t)nunits * sizeof(uint32_t);
    const size_t info_bytes = (size_t)nunits * BZ_IDS * sizeof(uint2);          // bz_link_kernel: two words per row
    if ((rc = semz_scratch(c, SEMZ_SLOT_BZ_AUX, segs_bytes + rows_bytes + ((ctl_bytes + 15) & ~(size_t)15) + info_bytes, &segs))) return rc;
    bz_row *rows = (bz_row *)((uint8_t *)segs + segs_bytes);
    uint32_t *ctl = (uint32_t *)((uint8_t *)rows + rows_bytes);
    uint2 *info = (uint2 *)((uint8_t *)ctl + ((ctl_bytes + 15) & ~(size_t)15));
    post = (uint8_t *)pre + (size_t)nunits * BZ_STRIDE;
    uint8_t *temp = (uint8_t *)tt + (size_t)nunits * BZ_STRIDE * sizeof(uint32_t);
    if (hipMemsetAsync(ctl, 0, ctl_bytes, c->stream) != hipSuccess) return semz_fail(c, -2, "hipMemsetAsync failed");
    uint32_t *starts = ctl, *extra = ctl + nunits;
    const bool narrow = !(c->bz_tables == 2 || (c->bz_tables == 0 && nunits <= BZ_WIDE_BLOCKS));
    auto decode = [&](uint32_t lo, uint32_t n, hipStream_t st) {
        const semz_unit *u = d_units + lo;
        bz_meta *m = (bz_meta *)meta + lo;
        uint8_t *pr = (uint8_t *)pre + (size_t)lo * BZ_STRIDE, *po = (uint8_t *)post + (size_t)lo * BZ_STRIDE;
        if (narrow) bz_decode_kernel<BZ_ROOT_NARROW><<<n, 128, 0, st>>>(d_in, in_len, u, n, out_cap, m, pr, po);
        else bz_decode_kernel<BZ_ROOT_WIDE><<<n, 128, 0, st>>>(d_in, in_len, u, n, out_cap, m, pr, po);
    };
    auto bwt = [&](uint32_t lo, uint32_t n, hipStream_t st, bool spans) {
        bz_meta *m = (bz_meta *)meta + lo;
        const uint8_t *pr = (const uint8_t *)pre + (size_t)lo * BZ_STRIDE;
        uint8_t *po = (uint8_t *)post + (size_t)lo * BZ_STRIDE;
        uint32_t *t = (uint32_t *)tt + (size_t)lo * BZ_STRIDE;
        uint8_t *tm = temp + (size_t)lo * BZ_TEMP_STRIDE;
        bz_row *rw = rows + (size_t)lo * BZ_IDS;
        if (spans) semz_span_begin(c, "bz_sort_kernel");
        bz_sort_kernel<<<n, BWT_THREADS, 0, st>>>(n, m, pr, t);
        if (spans) { semz_span_end(c); semz_span_begin(c, "bz_walk_kernel"); }
        bz_walk_kernel<<<BZ_XCDS * WALK_WGS, WALK_THREADS, 0, st>>>(n, m, t, tm, rw, starts + lo, extra + lo);
        if (spans) { semz_span_end(c); semz_span_begin(c, "bz_link_kernel"); }
        bz_link_kernel<<<n, BWT_THREADS, 0, st>>>(n, m, po, (bz_seg *)segs + (size_t)lo * BZ_SEGS, tm, rw, extra + lo, info + (size_t)lo * BZ_IDS);
        if (spans) semz_span_end(c);
    };
    // A batch that fills the chip several times over goes in parts: the decode of a part (bound by instruction issue)
    // runs beside the inverse BWT of the part before it (bound by the L2s), on a stream of its own.  Not with the
    // per-kernel timing on (the spans are events on one stream), and not for smaller batches, whose decode is the
    // latency of a block whatever their size (option "bzip2-overlap-blocks": the least blocks of a part, 2048).
    // Measured on 4800 blocks: 170.6 -> 163.8 ms; the halves decode in 44 ms each where the whole takes 80, and the
    // second half's inverse BWT has nothing to run beside, so the gain grows with the batch.
    const uint32_t per = c->bz_overlap_blocks;
    const uint32_t parts = c->profile || !per ? 1u : nunits / per < 4 ? nunits / per : 4u;
    if (parts < 2) {
        semz_span_begin(c, "bz_decode_kernel");
        decode(0, nunits, c->stream);
        semz_span_end(c);
        bwt(0, nunits, c->stream, true);
    } else {
        while (c->aux_events.size() < parts + 1) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return semz_fail(c, -2, "hipEventCreate failed");
            c->aux_events.push_back(e);
        }
        for (uint32_t p = 0; p < parts; p++) {
            const uint32_t lo = (uint32_t)((uint64_t)nunits * p / parts), hi = (uint32_t)((uint64_t)nunits * (p + 1) / parts);
            decode(lo, hi - lo, c->stream);
            if (hipEventRecord(c->aux_events[p], c->stream) != hipSuccess || hipStreamWaitEvent(c->aux_stream, c->aux_events[p], 0) != hipSuccess)
                return semz_fail(c, -2, "hipEventRecord / hipStreamWaitEvent failed");
            bwt(lo, hi - lo, c->aux_stream, false);
        }
        if (hipEventRecord(c->aux_events[parts], c->aux_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->aux_events[parts], 0) != hipSuccess)
            return semz_fail(c, -2, "hipEventRecord / hipStreamWaitEvent failed");
    }
    semz_span_begin(c, "bz_offsets_kernel");
    bz_offsets_kernel<<<1, OFFS_THREADS, 0, c->stream>>>(d_units, nunits, (bz_meta *)meta, out_cap, d_results, d_tails);
    semz_span_end(c);
    semz_span_begin(c, "bz_rle1_kernel");
    bz_rle1_kernel<<<nunits, 1024, 0, c->stream>>>(nunits, (const bz_meta *)meta, (const uint8_t *)post,
                                                   (const bz_seg *)segs, d_out);
    semz_span_end(c);
    semz_span_begin(c, "checksum_kernel");
    semz_launch_checksum(d_out, d_units, nunits, d_results, d_tails, c->d_tabs, 1u << SEMZ_TK_BZCRC, c->stream);
    semz_span_end(c);
    return 0;
}
