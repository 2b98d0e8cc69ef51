// Developer microbenchmark: cost of taken scalar branches, dependent SALU/VALU ops and cross-lane
// moves for a single wave per SIMD on gfx950 (the regime the decode kernels run in).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k_branch(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_branch 0\n" ::: "memory");)      // branch to the next instruction: taken
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_cbranch_nt(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 0\n" ::: "memory", "scc");)   // never taken
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_salu(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = iters;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_add_u32 %0, %0, 3\n" : "+s"(x) : : "scc");)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (x == 12345);
}
__global__ void k_valu(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("v_add_u32 %0, %0, 3\n" : "+v"(x));)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345) out[1000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_readlane(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x, s = 0;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n" : "+s"(s) : "v"(x) : "scc");)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (s == 12345);
}
__global__ void k_dpp(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n" : "+v"(x));)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345) out[1000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(unsigned long long *out, int iters)
{
    __shared__ unsigned tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = tab[x];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345) out[1000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void k_lds_rfl(unsigned long long *out, int iters)
{
    __shared__ unsigned tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = 5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = __builtin_amdgcn_readfirstlane(tab[x]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345) out[1000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
// a mixed scalar-heavy body (the shape of the decode loops): how does the time grow with waves per SIMD?
__global__ void k_mix(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = threadIdx.x, s = 0, a = iters;
    for (int i = 0; i < iters; i++) {
        REP16(asm volatile("v_readlane_b32 %0, %2, %0\n s_and_b32 %0, %0, 63\n s_add_u32 %1, %1, %0\n s_lshr_b32 %1, %1, 1\n v_add_u32 %2, %2, %1\n s_cmp_eq_u32 %1, 0x7fffffff\n s_cbranch_scc1 0\n v_xor_b32 %2, %2, %1\n"
                           : "+s"(s), "+s"(a), "+v"(x) : : "scc");)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345) out[1000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x & 1023] = t1 - t0 + (s == 12345);
}
int main()
{
    unsigned long long *d, h[1];
    hipMalloc(&d, 8192);
    const int iters = 100000;
    const int grids[2] = {1, 1024};         // one wave alone; one wave on every SIMD
    for (int gi = 0; gi < 2; gi++) {
        int g = grids[gi];
#define RUN(k, what) do { k<<<g, 64>>>(d, iters); hipDeviceSynchronize(); k<<<g, 64>>>(d, iters); hipDeviceSynchronize(); \
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("grid %4d %-34s %7.1f clocks each\n", g, what, (double)h[0] / (16.0 * iters)); fflush(stdout); } while (0)
        RUN(k_branch, "taken s_branch");
        RUN(k_cbranch_nt, "s_cmp + untaken s_cbranch");
        RUN(k_salu, "dependent s_add");
        RUN(k_valu, "dependent v_add");
        RUN(k_readlane, "readlane + s_and (dependent)");
        RUN(k_dpp, "s_nop 1 + dpp wave_shr (dependent)");
        RUN(k_lds, "dependent ds_read per lane");
        RUN(k_lds_rfl, "ds_read + readfirstlane (dependent)");
    }
    for (int g = 256; g <= 16384; g *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_mix<<<g, 64>>>(d, 20000); hipDeviceSynchronize();
        hipEventRecord(e0); k_mix<<<g, 64>>>(d, 20000); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("mixed body: %5d waves (%.2f per SIMD)  %8.3f ms  %6.1f ticks per 8-instruction group in wave 0\n", g, g / 1024.0, ms, (double)h[0] / (16.0 * 20000));
        fflush(stdout);
    }
    return 0;
}
