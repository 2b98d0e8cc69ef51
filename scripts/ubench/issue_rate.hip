// Developer microbenchmark: issue rate of one wave for independent scalar / vector instructions and for the
// LZMA bit step as lzma2_fast.hip writes it (gfx950).  Ticks are s_memtime ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define T0 unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define T1(extra) unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[blockIdx.x & 1023] = t1 - t0 + (extra);
__global__ void k_salu_dep(unsigned long long *out, int iters)
{
    T0 unsigned x = iters;
    for (int i = 0; i < iters; i++) { REP16(asm volatile("s_add_u32 %0, %0, 3\n" : "+s"(x) : : "scc");) }
    T1(x == 12345)
}
__global__ void k_salu_indep(unsigned long long *out, int iters)
{
    T0 unsigned a = iters, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    for (int i = 0; i < iters; i++) {
        asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                     "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n"
                     "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                     "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : : "scc");
    }
    T1((a + b + c + d + e + f + g + h) == 12345)
}
__global__ void k_smov_indep(unsigned long long *out, int iters)
{
    T0 unsigned a = iters, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    for (int i = 0; i < iters; i++) {
        asm volatile("s_mov_b32 %0, 3\n s_mov_b32 %1, 3\n s_mov_b32 %2, 3\n s_mov_b32 %3, 3\n"
                     "s_mov_b32 %4, 3\n s_mov_b32 %5, 3\n s_mov_b32 %6, 3\n s_mov_b32 %7, 3\n"
                     "s_mov_b32 %0, 4\n s_mov_b32 %1, 4\n s_mov_b32 %2, 4\n s_mov_b32 %3, 4\n"
                     "s_mov_b32 %4, 4\n s_mov_b32 %5, 4\n s_mov_b32 %6, 4\n s_mov_b32 %7, 4\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : :);
    }
    T1((a + b + c + d + e + f + g + h) == 12345)
}
__global__ void k_valu_indep(unsigned long long *out, int iters)
{
    T0 unsigned a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 3\n v_add_u32 %3, %3, 3\n"
                     "v_add_u32 %4, %4, 3\n v_add_u32 %5, %5, 3\n v_add_u32 %6, %6, 3\n v_add_u32 %7, %7, 3\n"
                     "v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 3\n v_add_u32 %3, %3, 3\n"
                     "v_add_u32 %4, %4, 3\n v_add_u32 %5, %5, 3\n v_add_u32 %6, %6, 3\n v_add_u32 %7, %7, 3\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    }
    if ((a + b + c + d + e + f + g + h) == 12345) out[1000] = 1;
    T1(0)
}
__global__ void k_mixed_indep(unsigned long long *out, int iters)
{
    T0 unsigned a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_u32 %0, %0, 3\n s_add_u32 %4, %4, 3\n v_add_u32 %1, %1, 3\n s_add_u32 %5, %5, 3\n"
                     "v_add_u32 %2, %2, 3\n s_add_u32 %6, %6, 3\n v_add_u32 %3, %3, 3\n s_add_u32 %7, %7, 3\n"
                     "v_add_u32 %0, %0, 3\n s_add_u32 %4, %4, 3\n v_add_u32 %1, %1, 3\n s_add_u32 %5, %5, 3\n"
                     "v_add_u32 %2, %2, 3\n s_add_u32 %6, %6, 3\n v_add_u32 %3, %3, 3\n s_add_u32 %7, %7, 3\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : : "scc");
    }
    if ((a + b + c + d) == 12345) out[1000] = 1;
    T1((e + f + g + h) == 12345)
}
// the bit step of lzma2_fast.hip, 16 per loop pass; the normalisation is never taken here (range is reset)
#define BITSTEP \
    "s_mov_b32 m0, %[q]\n v_readlane_b32 %[p], %[vt], m0\n s_lshr_b32 %[b], %[range], 11\n s_mul_i32 %[b], %[b], %[p]\n" \
    "s_sub_u32 %[nr], %[range], %[b]\n s_sub_u32 %[nc], %[code], %[b]\n s_cselect_b32 %[range], %[b], %[nr]\n"      \
    "s_cselect_b32 %[code], %[code], %[nc]\n s_cselect_b32 %[k], 0x7e1, 0\n s_addc_u32 %[q], %[q], %[q]\n"           \
    "s_sub_i32 %[k], %[p], %[k]\n s_ashr_i32 %[k], %[k], 5\n s_sub_i32 %[p], %[p], %[k]\n v_writelane_b32 %[vt], %[p], m0\n" \
    "s_and_b32 %[q], %[q], 63\n s_or_b32 %[range], %[range], 0x80000000\n s_cmp_gt_u32 %[range], 0xffffff\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"
__global__ void k_bitstep(unsigned long long *out, int iters)
{
    T0 unsigned vt = 1024, range = 0xffffffffu, code = iters * 2654435761u, q = 1, p, b, nr, nc, k;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile(BITSTEP : [vt] "+v"(vt), [range] "+s"(range), [code] "+s"(code), [q] "+s"(q), [p] "=&s"(p), [b] "=&s"(b),
                           [nr] "=&s"(nr), [nc] "=&s"(nc), [k] "=&s"(k) : : "scc", "m0");)
    }
    if (vt == 12345) out[1000] = 1;
    T1((range ^ code ^ q) == 12345)
}
// the same with the probability update moved to the vector pipe for all lanes at once (mask = one lane)
#define BITSTEP_V \
    "s_mov_b32 m0, %[q]\n v_readlane_b32 %[p], %[vt], m0\n s_lshr_b32 %[b], %[range], 11\n s_mul_i32 %[b], %[b], %[p]\n" \
    "s_sub_u32 %[nr], %[range], %[b]\n s_sub_u32 %[nc], %[code], %[b]\n s_cselect_b32 %[range], %[b], %[nr]\n"      \
    "s_cselect_b32 %[code], %[code], %[nc]\n s_cselect_b32 %[k], 0x7e1, 0\n s_addc_u32 %[q], %[q], %[q]\n"           \
    "s_lshl_b64 vcc, 1, m0\n v_sub_u32 %[t], %[vt], %[k]\n v_ashrrev_i32 %[t], 5, %[t]\n v_sub_u32 %[t], %[vt], %[t]\n v_cndmask_b32 %[vt], %[vt], %[t], vcc\n" \
    "s_and_b32 %[q], %[q], 63\n s_or_b32 %[range], %[range], 0x80000000\n s_cmp_gt_u32 %[range], 0xffffff\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"
__global__ void k_bitstep_v(unsigned long long *out, int iters)
{
    T0 unsigned vt = 1024, range = 0xffffffffu, code = iters * 2654435761u, q = 1, p, b, nr, nc, k, t;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile(BITSTEP_V : [vt] "+v"(vt), [range] "+s"(range), [code] "+s"(code), [q] "+s"(q), [p] "=&s"(p), [b] "=&s"(b),
                           [nr] "=&s"(nr), [nc] "=&s"(nc), [k] "=&s"(k), [t] "=&v"(t) : : "scc", "m0", "vcc");)
    }
    if (vt == 12345) out[1000] = 1;
    T1((range ^ code ^ q) == 12345)
}
int main()
{
    unsigned long long *d, h[1];
    hipMalloc(&d, 16384);
    const int iters = 50000;
    const int grids[5] = {1, 1024, 2048, 4096, 8192};   // one wave alone; one, two, four, eight waves on every SIMD
    for (int gi = 0; gi < 5; gi++) {
        int g = grids[gi];
#define RUN(k, what, n) do { k<<<g, 64>>>(d, iters); hipDeviceSynchronize(); k<<<g, 64>>>(d, iters); hipDeviceSynchronize(); \
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("grid %4d %-44s %7.2f ticks per instruction\n", g, what, (double)h[0] / (16.0 * iters * (n))); fflush(stdout); } while (0)
        RUN(k_salu_dep, "dependent s_add", 1);
        RUN(k_salu_indep, "independent s_add (8 chains)", 1);
        RUN(k_smov_indep, "independent s_mov", 1);
        RUN(k_valu_indep, "independent v_add (8 chains)", 1);
        RUN(k_mixed_indep, "alternating independent v_add / s_add", 1);
        RUN(k_bitstep, "bit step, scalar update (19 instructions)", 19);
        RUN(k_bitstep_v, "bit step, vector update (20 instructions)", 20);
    }
    return 0;
}
