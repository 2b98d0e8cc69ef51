// checksums.hip -- gfx950 kernels for the checksums of the reference and the
// per-unit trailer verification.
//
//   CRC-32        update-crc-32   crc.lisp:19-110      (gzip trailer, xz check type 1)
//   CRC-64/XZ     update-crc-64   crc.lisp:122-193     (xz check type 4)
//   Adler-32      update-adler-32 adler32.lisp:54-87   (zlib trailer)
//   bzip2 CRC     update-bzip2-crc bzip2.lisp:13-39    (block CRC, MSB-first 0x04C11DB7)
//   SHA-256       sha256.lisp:27-164                   (xz check type 10)
//
// One wave per unit.  The unit's output is cut into 64 lane segments whose
// starts are 16-byte aligned in HBM; each lane runs a table-driven CRC over
// its segment (slice-by-4, tables in LDS) from state 0 (lane 0: the real init
// value), and the lane states are folded with GF(2) polynomial multiplication:
//   crc(A || B) = crc(A) * x^(8|B|) mod P  xor  crc0(B)
// Adler-32 folds with s2 = sum(s2_i + s1_i * bytes_after_i).  SHA-256 has no
// such fold (Merkle-Damgard): it runs one unit per lane.
// The last lane then applies the reference's trailer checks in the
// reference's order (gzip.lisp:102-113, zlib.lisp:66-76, xz.lisp:444-454).
#include "semz_host.h"


namespace {

// a*b mod P in the reflected representation (bit 31 = x^0), zlib's multmodp
__device__ uint32_t mulmod32r(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & m) p ^= b;
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__device__ uint64_t mulmod64r(uint64_t a, uint64_t b)
{
    uint64_t m = 1ull << 63, p = 0;
    for (int i = 0; i < 64; i++) {
        if (a & m) p ^= b;
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xC96C5795D7870F42ull : b >> 1;
    }
    return p;
}
// a*b mod P, natural representation (bit k = x^k), P = x^32 + 0x04C11DB7
__device__ uint32_t mulmod32n(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1;
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}

// state * x^(8*nbytes) mod P
__device__ uint32_t shift32r(const uint32_t *x2n, uint32_t st, uint64_t nbytes)
{
    for (int k = 3; nbytes; nbytes >>= 1, k++)
        if (nbytes & 1) st = mulmod32r(x2n[k & 31], st);
    return st;
}
__device__ uint32_t shift32n(const uint32_t *x2n, uint32_t st, uint64_t nbytes)
{
    for (int k = 3; nbytes; nbytes >>= 1, k++)
        if (nbytes & 1) st = mulmod32n(x2n[k & 31], st);
    return st;
}
__device__ uint64_t shift64r(const uint64_t *x2n, uint64_t st, uint64_t nbytes)
{
    for (int k = 3; nbytes; nbytes >>= 1, k++)
        if (nbytes & 1) st = mulmod64r(x2n[k & 63], st);
    return st;
}

__device__ __forceinline__ uint32_t xor_reduce32(uint32_t v)
{
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t xor_reduce64(uint64_t v)
{
    uint32_t lo = xor_reduce32((uint32_t)v), hi = xor_reduce32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t add_reduce64(uint64_t v)
{
    for (int o = 32; o; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

struct Seg { uint64_t beg, end; };
// lane segment [beg,end) of a unit of n bytes whose first byte has address
// misalignment a (0..15): all segment starts but the first are 16-byte aligned
__device__ __forceinline__ Seg lane_segment(uint64_t n, uint32_t a, int lane)
{
    uint64_t s = ((n + a + 63) / 64 + 15) & ~15ull;
    uint64_t b = (uint64_t)lane * s, e = b + s;
    Seg g;
    g.beg = b > a ? b - a : 0;
    g.end = e > a ? e - a : 0;
    if (g.beg > n) g.beg = n;
    if (g.end > n) g.end = n;
    return g;
}

constexpr int CK_WAVES = 4;

__global__ __launch_bounds__(64 * CK_WAVES) void checksum_kernel(const uint8_t *out, const semz_unit *units,
                                                                 uint32_t nunits, semz_result *results,
                                                                 const semz_tail *tails,
                                                                 const semz_check_tables *tabs, uint32_t kind_mask)
{
    __shared__ uint32_t t32[4][256];
    __shared__ uint64_t t64[4][256];
    const int lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x * CK_WAVES + (threadIdx.x >> 6);
    // stage the tables this launch needs
    if (kind_mask & ((1u << SEMZ_TK_CRC32) | (1u << SEMZ_TK_BZCRC))) {
        const uint32_t *src = (kind_mask & (1u << SEMZ_TK_CRC32)) ? &tabs->crc32[0][0] : &tabs->bzcrc[0][0];
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) (&t32[0][0])[i] = src[i];
    }
    if (kind_mask & (1u << SEMZ_TK_CRC64))
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) (&t64[0][0])[i] = (&tabs->crc64[0][0])[i];
    __syncthreads();
    if (u >= nunits) return;

    semz_result r = results[u];
    const semz_tail t = tails[u];
    if (r.status != SEMZ_OK) return;
    const uint8_t *p = out + r.out_off;
    const uint64_t n = r.out_len;
    const uint32_t a = (uint32_t)((uintptr_t)p & 15);
    const Seg g = lane_segment(n, a, lane);
    const uint64_t after = n - g.end;

    uint32_t check[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t check_len = 0;

    if (t.check_kind == SEMZ_TK_CRC32) {
        uint32_t c = (g.beg == 0 && lane == 0) ? 0xFFFFFFFFu : 0u;
        uint64_t i = g.beg;
        for (; i < g.end && ((uintptr_t)(p + i) & 15); i++) c = t32[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
        // (64 bytes per lane and trip: the lanes of a wave read 64 different lines, and a line should not have to
        // wait in the cache for eight trips until its lane is through with it)
        for (; i + 64 <= g.end; i += 64) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p + i);
            const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                c ^= w[k];
                c = t32[3][c & 0xFF] ^ t32[2][(c >> 8) & 0xFF] ^ t32[1][(c >> 16) & 0xFF] ^ t32[0][c >> 24];
            }
        }
        for (; i + 16 <= g.end; i += 16) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + i);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c ^= w[k];
                c = t32[3][c & 0xFF] ^ t32[2][(c >> 8) & 0xFF] ^ t32[1][(c >> 16) & 0xFF] ^ t32[0][c >> 24];
            }
        }
        for (; i < g.end; i++) c = t32[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
        c = shift32r(tabs->x2n32, c, after);
        check[0] = ~xor_reduce32(c);
        check_len = 4;
    } else if (t.check_kind == SEMZ_TK_BZCRC) {
        uint32_t c = (g.beg == 0 && lane == 0) ? 0xFFFFFFFFu : 0u;
        uint64_t i = g.beg;
        for (; i < g.end && ((uintptr_t)(p + i) & 15); i++) c = (c << 8) ^ t32[0][(c >> 24) ^ p[i]];
        for (; i + 64 <= g.end; i += 64) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p + i);
            const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                c ^= __builtin_bswap32(w[k]);
                c = t32[3][c >> 24] ^ t32[2][(c >> 16) & 0xFF] ^ t32[1][(c >> 8) & 0xFF] ^ t32[0][c & 0xFF];
            }
        }
        for (; i + 16 <= g.end; i += 16) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + i);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c ^= __builtin_bswap32(w[k]);
                c = t32[3][c >> 24] ^ t32[2][(c >> 16) & 0xFF] ^ t32[1][(c >> 8) & 0xFF] ^ t32[0][c & 0xFF];
            }
        }
        for (; i < g.end; i++) c = (c << 8) ^ t32[0][(c >> 24) ^ p[i]];
        c = shift32n(tabs->x2nbz, c, after);
        check[0] = ~xor_reduce32(c);
        check_len = 4;
    } else if (t.check_kind == SEMZ_TK_CRC64) {
        uint64_t c = (g.beg == 0 && lane == 0) ? ~0ull : 0ull;
        uint64_t i = g.beg;
        for (; i < g.end && ((uintptr_t)(p + i) & 15); i++) c = t64[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
        for (; i + 64 <= g.end; i += 64) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p + i);
            const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                c ^= w[k];
                c = t64[3][c & 0xFF] ^ t64[2][(c >> 8) & 0xFF] ^ t64[1][(c >> 16) & 0xFF] ^
                    t64[0][(c >> 24) & 0xFF] ^ (c >> 32);
            }
        }
        for (; i + 16 <= g.end; i += 16) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + i);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c ^= w[k];
                c = t64[3][c & 0xFF] ^ t64[2][(c >> 8) & 0xFF] ^ t64[1][(c >> 16) & 0xFF] ^
                    t64[0][(c >> 24) & 0xFF] ^ (c >> 32);
            }
        }
        for (; i < g.end; i++) c = t64[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
        c = shift64r(tabs->x2n64, c, after);
        c = ~xor_reduce64(c);
        check[0] = (uint32_t)c;
        check[1] = (uint32_t)(c >> 32);
        check_len = 8;
    } else if (t.check_kind == SEMZ_TK_ADLER32) {
        uint32_t s1 = 0, s2 = 0;
        uint64_t i = g.beg;
        while (i < g.end) {
            uint64_t stop = g.end - i > 5552 ? i + 5552 : g.end;   // deferred modulo, adler32.lisp:63-70
            for (; i < stop && ((uintptr_t)(p + i) & 15); i++) { s1 += p[i]; s2 += s1; }
            for (; i + 64 <= stop; i += 64) {
                const uint4 *q = reinterpret_cast<const uint4 *>(p + i);
                const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
                const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    s1 += w[k] & 0xFF; s2 += s1;
                    s1 += (w[k] >> 8) & 0xFF; s2 += s1;
                    s1 += (w[k] >> 16) & 0xFF; s2 += s1;
                    s1 += w[k] >> 24; s2 += s1;
                }
            }
            for (; i + 16 <= stop; i += 16) {
                uint4 v = *reinterpret_cast<const uint4 *>(p + i);
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s1 += w[k] & 0xFF; s2 += s1;
                    s1 += (w[k] >> 8) & 0xFF; s2 += s1;
                    s1 += (w[k] >> 16) & 0xFF; s2 += s1;
                    s1 += w[k] >> 24; s2 += s1;
                }
            }
            for (; i < stop; i++) { s1 += p[i]; s2 += s1; }
            s1 %= 65521u;
            s2 %= 65521u;
        }
        uint64_t part2 = ((uint64_t)s2 + (uint64_t)s1 * (after % 65521u)) % 65521u;
        uint64_t S1 = (1 + add_reduce64(s1)) % 65521u;
        uint64_t S2 = (n % 65521u + add_reduce64(part2)) % 65521u;
        check[0] = (uint32_t)((S2 << 16) | S1);
        check_len = 4;
    } else if (t.check_kind == SEMZ_TK_SHA256) {
        // computed by sha256_kernel, launched in front of this one: it left the digest in the result
        for (int k = 0; k < 8; k++)
            check[k] = (uint32_t)r.check[4 * k] | (uint32_t)r.check[4 * k + 1] << 8 | (uint32_t)r.check[4 * k + 2] << 16 |
                       (uint32_t)r.check[4 * k + 3] << 24;
        check_len = 32;
    }

    if (lane == 0) {
        for (int k = 0; k < 8; k++) {
            uint32_t w = check[k];
            r.check[4 * k] = (uint8_t)w; r.check[4 * k + 1] = (uint8_t)(w >> 8);
            r.check[4 * k + 2] = (uint8_t)(w >> 16); r.check[4 * k + 3] = (uint8_t)(w >> 24);
        }
        r.check_len = check_len;
        bool match = true;
        for (uint32_t k = 0; k < (check_len + 3) / 4; k++) match = match && check[k] == t.want[k];
        switch (t.check_kind) {
        case SEMZ_TK_CRC32:
            if (units[u].kind == SEMZ_UNIT_GZIP) {              // gzip.lisp:102-113
                if (!t.have_check) r.status = SEMZ_EOF;
                else if (!match) { r.status = SEMZ_E_GZIP_DATA_CRC; r.detail = t.want[0]; }
                else if (!t.have_isize) r.status = SEMZ_EOF;
                else if (t.want_isize != (uint32_t)n) { r.status = SEMZ_E_GZIP_ISIZE; r.detail = t.want_isize; }
                break;
            }
            /* fallthrough: xz block check */
        case SEMZ_TK_CRC64:
        case SEMZ_TK_SHA256:                                    // xz.lisp:444-454
            if (!t.have_check) r.status = SEMZ_EOF;
            else if (!match) { r.status = SEMZ_E_XZ_BLOCK_CHECK; r.detail = t.want[0]; }
            break;
        case SEMZ_TK_ADLER32:                                   // zlib.lisp:66-76
            if (!t.have_check) r.status = SEMZ_EOF;
            else if (!match) { r.status = SEMZ_E_ZLIB_CHECKSUM; r.detail = t.want[0]; }
            break;
        case SEMZ_TK_BZCRC:                                     // bzip2.lisp:517-521
            if (!match) { r.status = SEMZ_E_BZ_BLOCK_CRC; r.detail = t.want[0]; }
            break;
        default: break;
        }
        results[u] = r;
    }
}

// SHA-256 (RFC 6234): sha256.lisp:27-164; constants tables.lisp:152-197.
// A unit's digest is one serial chain of compressions (16,384 for a 1 MiB block), so a unit is one wave and the
// chain is kept as short as it can be: the message schedule does not belong to it.  Per pass of 64 message blocks,
// lane b loads block b (a 64-byte line), expands it to the 64 schedule words, adds the round constants and leaves
// them in LDS; then all lanes run the 64 x 64 rounds on the same values (sixteen vector instructions a round,
// the W + K words coming as LDS broadcasts, four at a time).  One lane doing everything took 254 ms for 2048
// blocks of 1 MiB, 42 % of config 4.
__constant__ uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

constexpr int SHA_STRIDE = 68;                 // words per block in LDS: 64 and a pad that keeps 16-byte alignment

__global__ __launch_bounds__(64) void sha256_kernel(const uint8_t *out, uint32_t nunits, semz_result *results, const semz_tail *tails)
{
    __shared__ __align__(16) uint32_t wk[64 * SHA_STRIDE];
    const uint32_t u = blockIdx.x, lane = threadIdx.x;
    if (u >= nunits) return;
    if (results[u].status != SEMZ_OK || tails[u].check_kind != SEMZ_TK_SHA256) return;
    const uint8_t *p = out + results[u].out_off;
    const uint64_t n = results[u].out_len;
    const uint64_t nblocks = (n + 9 + 63) / 64;                    // data, 0x80, zeros, 64-bit length
    uint32_t h0 = 0x6a09e667, h1 = 0xbb67ae85, h2 = 0x3c6ef372, h3 = 0xa54ff53a, h4 = 0x510e527f, h5 = 0x9b05688c,
             h6 = 0x1f83d9ab, h7 = 0x5be0cd19;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 64) {
        const uint64_t j = b0 + lane;
        const uint32_t nb = nblocks - b0 < 64 ? (uint32_t)(nblocks - b0) : 64u;
        if (lane < nb) {
            uint32_t w[16];
            if (64 * j + 64 <= n) {
                uint4 q[4];
                __builtin_memcpy(q, p + 64 * j, 64);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    w[4 * k] = __builtin_bswap32(q[k].x); w[4 * k + 1] = __builtin_bswap32(q[k].y);
                    w[4 * k + 2] = __builtin_bswap32(q[k].z); w[4 * k + 3] = __builtin_bswap32(q[k].w);
                }
            } else {
                // one of the last two blocks: the rest of the data, the 0x80, zeros, and the bit count at the very end
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    uint32_t x = 0;
                    for (int q = 0; q < 4; q++) {
                        const uint64_t at = 64 * j + 4 * k + q;
                        const uint32_t byte = at < n ? p[at] : at == n ? 0x80u : 0u;
                        x = (x << 8) | byte;
                    }
                    w[k] = x;
                }
                if (j + 1 == nblocks) { w[14] = (uint32_t)((n * 8) >> 32); w[15] = (uint32_t)(n * 8); }
            }
            uint32_t *dst = wk + lane * SHA_STRIDE;
#pragma unroll
            for (int t = 0; t < 64; t++) {
                if (t >= 16) {
                    const uint32_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                    const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
                    const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
                    w[t & 15] = s1 + w[(t - 7) & 15] + s0 + w[t & 15];
                }
                dst[t] = w[t & 15] + K256[t];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the chain: every lane computes the same values
        for (uint32_t b = 0; b < nb; b++) {
            const uint4 *src = reinterpret_cast<const uint4 *>(wk + b * SHA_STRIDE);
            uint32_t a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, g = h6, hh = h7;
            // (the values are the same in every lane and the compiler knows it: left alone it moves the chain to
            // the scalar unit, which has no rotate, and pays a v_alignbit + v_readfirstlane round trip for each;
            // the empty asm statements pin the chain to the vector pipe)
            asm volatile("" : "+v"(a), "+v"(bb), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(hh));
#pragma unroll
            for (int t4 = 0; t4 < 16; t4++) {
                uint4 x = src[t4];
                asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));
                const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) | (~e & g)) + xs[q];
                    const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) | (c & (a | bb)));
                    hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
                }
            }
            h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += hh;
            asm volatile("" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (lane == 0) {
        // digest bytes = big-endian words (sha256.lisp:159-164)
        const uint32_t hw[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
        semz_result *r = results + u;
        for (int k = 0; k < 8; k++) {
            r->check[4 * k] = (uint8_t)(hw[k] >> 24); r->check[4 * k + 1] = (uint8_t)(hw[k] >> 16);
            r->check[4 * k + 2] = (uint8_t)(hw[k] >> 8); r->check[4 * k + 3] = (uint8_t)hw[k];
        }
    }
}

} // namespace

extern "C" void semz_launch_checksum(const uint8_t *d_out, const semz_unit *d_units, uint32_t nunits,
                                     semz_result *d_results, const semz_tail *d_tails,
                                     const semz_check_tables *d_tabs, uint32_t kind_mask, hipStream_t stream)
{
    if (nunits == 0) return;
    if (kind_mask & (1u << SEMZ_TK_SHA256)) sha256_kernel<<<nunits, 64, 0, stream>>>(d_out, nunits, d_results, d_tails);
    checksum_kernel<<<(nunits + CK_WAVES - 1) / CK_WAVES, 64 * CK_WAVES, 0, stream>>>(
        d_out, d_units, nunits, d_results, d_tails, d_tabs, kind_mask);
}

// Host-side construction of the tables (called once per context).
extern "C" void semz_build_check_tables(semz_check_tables *t)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;                                   // crc.lisp:24-30
        for (int j = 0; j < 8; j++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        t->crc32[0][i] = c;
        uint64_t d = i;                                   // crc.lisp:127-133
        for (int j = 0; j < 8; j++) d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
        t->crc64[0][i] = d;
        uint32_t b = i << 24;                             // tables.lisp:10-82 (bzip2 crctable)
        for (int j = 0; j < 8; j++) b = (b & 0x80000000u) ? (b << 1) ^ 0x04C11DB7u : b << 1;
        t->bzcrc[0][i] = b;
    }
    for (int k = 1; k < 4; k++)
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = t->crc32[k - 1][i];
            t->crc32[k][i] = (c >> 8) ^ t->crc32[0][c & 0xFF];
            uint64_t d = t->crc64[k - 1][i];
            t->crc64[k][i] = (d >> 8) ^ t->crc64[0][d & 0xFF];
            uint32_t b = t->bzcrc[k - 1][i];
            t->bzcrc[k][i] = (b << 8) ^ t->bzcrc[0][b >> 24];
        }
    // x^(2^k) mod P by repeated squaring (host versions of the mulmods)
    auto mm32r = [](uint32_t a, uint32_t b) {
        uint32_t m = 1u << 31, p = 0;
        for (int i = 0; i < 32; i++) { if (a & m) p ^= b; m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
        return p;
    };
    auto mm64r = [](uint64_t a, uint64_t b) {
        uint64_t m = 1ull << 63, p = 0;
        for (int i = 0; i < 64; i++) { if (a & m) p ^= b; m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xC96C5795D7870F42ull : b >> 1; }
        return p;
    };
    auto mm32n = [](uint32_t a, uint32_t b) {
        uint32_t r = 0;
        for (int i = 31; i >= 0; i--) { r = (r & 0x80000000u) ? (r << 1) ^ 0x04C11DB7u : r << 1; if ((b >> i) & 1) r ^= a; }
        return r;
    };
    uint32_t p = 1u << 30;                                // x^1 reflected
    for (int k = 0; k < 32; k++) { t->x2n32[k] = p; p = mm32r(p, p); }
    uint64_t q = 1ull << 62;
    for (int k = 0; k < 64; k++) { t->x2n64[k] = q; q = mm64r(q, q); }
    uint32_t s = 2;                                       // x^1 natural
    for (int k = 0; k < 32; k++) { t->x2nbz[k] = s; s = mm32n(s, s); }
}
