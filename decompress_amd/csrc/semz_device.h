// semz_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Execution model of the decode kernels: a unit (deflate member, bzip2 block,
// LZMA2 block) is decoded by one 64-lane wavefront (bzip2: two) whose control
// state is wave-uniform; the lanes run the data-parallel parts (speculative
// decode at every bit offset of a window, table construction, match copies,
// ring flushes).  Inside one wave LDS hand-offs between lanes need no
// s_barrier: LDS operations of one wave execute in program order, and
// lds_sync() only stops the compiler from moving LDS accesses across it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/semz_gpu.h"

#define SEMZ_WAVE 64

// What the decode kernels leave for the checksum/verify kernels.
struct semz_tail {
    uint32_t want[8];     // expected check value (little-endian words) read from the trailer
    uint32_t have_check;  // 1 = the check field was present (0 = input ended first -> EOF)
    uint32_t have_isize;  // gzip: ISIZE field present
    uint32_t want_isize;
    uint32_t check_kind;  // SEMZ_TK_*
    uint64_t trailer_end_bit; // in_end_bit if every trailer field was present
    uint64_t pad;
};
enum { SEMZ_TK_NONE = 0, SEMZ_TK_CRC32 = 1, SEMZ_TK_ADLER32 = 2, SEMZ_TK_BZCRC = 3,
       SEMZ_TK_CRC64 = 4, SEMZ_TK_SHA256 = 5 };

// Span checks of the device entry points (semz_decode_batch validates on the host and fails the call; the device
// entry points cannot see the unit table, so every decode kernel starts with this and reports SEMZ_E_BAD_UNIT
// for a unit whose spans leave the buffers of the call, without reading or writing anything through them).
// family: 0 deflate/zlib/gzip, 1 bzip2 blocks, 2 xz blocks / LZMA2 / LZMA.
__device__ __forceinline__ bool semz_unit_spans_ok(const semz_unit &un, uint64_t in_len, uint64_t out_cap, int family)
{
    const int fam = un.kind <= SEMZ_UNIT_GZIP ? 0 : un.kind == SEMZ_UNIT_BZIP2_BLOCK ? 1 : un.kind <= SEMZ_UNIT_LZMA_WIDE ? 2 : 3;
    bool ok = fam == family && un.in_off <= un.in_limit && un.in_limit <= in_len && un.in_bit_off <= 7;
    if (un.out_off == SEMZ_OUT_AUTO) ok = ok && family == 1;
    else ok = ok && un.out_off <= out_cap && un.out_cap <= out_cap - un.out_off;
    if (un.prefix_len) ok = ok && un.prefix_off <= in_len && un.prefix_len <= in_len - un.prefix_off;
    return ok;
}

// The arbiter of a SIMD takes the oldest wave that is ready: of the waves that share a SIMD one runs ahead and the
// last one ends alone, on a fraction of the SIMD's issue rate.  Long-running decoders call this now and then: the
// waves take turns at the four priorities, by the constant 100 MHz clock and their slot on the SIMD, and end
// together (inflate_kernel: 9.94 -> 8.81 ms for config 2; scripts/inflate_unit_times.py shows the finish times).
// A turn lasts 2^shift ticks.
__device__ __forceinline__ void semz_take_turn(int shift)
{
    const uint32_t slot = __builtin_amdgcn_s_getreg((4 << 11) | 4);              // HW_ID: wave slot, bits 3:0
    switch (((uint32_t)(__builtin_amdgcn_s_memrealtime() >> shift) + slot) & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int G> struct Group {
    static_assert(G == 64 || G == 32 || G == 16 || G == 8 || G == 4, "group size");
    __device__ __forceinline__ static int lane() { return threadIdx.x & 63; }
    __device__ __forceinline__ static int gl() { return lane() & (G - 1); }
    __device__ __forceinline__ static int base() { return lane() & ~(G - 1); }
    // value held by the group's leader
    template <typename T> __device__ __forceinline__ static T bcast(T v)
    {
        if constexpr (sizeof(T) == 8) {
            uint32_t lo = (uint32_t)(uint64_t)v, hi = (uint32_t)((uint64_t)v >> 32);
            lo = bcast(lo);
            hi = bcast(hi);
            return (T)(((uint64_t)hi << 32) | lo);
        } else if constexpr (G == 64) {
            return (T)__builtin_amdgcn_readfirstlane((int)v);
        } else {
            return (T)__shfl((int)v, base(), 64);
        }
    }
    __device__ __forceinline__ static uint64_t ballot(bool p)
    {
        uint64_t m = __ballot(p);
        if constexpr (G == 64) return m;
        else return (m >> base()) & ((1ull << G) - 1);
    }
};

// unaligned little-endian loads from global memory
__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// bytes [pos, pos+8) of a buffer that ends at `limit`; bytes beyond read as zero
__device__ __forceinline__ uint64_t ld_u64_clamped(const uint8_t *base, uint64_t pos, uint64_t limit)
{
    if (pos + 8 <= limit) return ld_u64(base + pos);
    uint64_t v = 0;
    for (int i = 0; i < 8; i++)
        if (pos + i < limit) v |= (uint64_t)base[pos + i] << (8 * i);
    return v;
}
