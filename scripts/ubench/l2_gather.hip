// Developer microbenchmark: does a table of T bytes gathered at random by the workgroups of ONE XCD stay in that
// XCD's L2?  Every workgroup reads its XCC_ID (hardware register 20) and walks dependent random chains through the
// table that belongs to that id; time per step against T, and the XCC_ID of the first workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <numeric>
#include <random>
#include <algorithm>
__global__ void k_ids(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }
__global__ __launch_bounds__(512) void k_walk(const uint32_t *tabs, uint32_t entries, uint32_t steps, uint32_t *sink, int use_xcc)
{
    const uint32_t xcd = use_xcc ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) % 8 : blockIdx.x % 8;
    const uint32_t *t = tabs + (size_t)xcd * entries;
    uint32_t k = (blockIdx.x * 512 + threadIdx.x) * 977u % entries;
    for (uint32_t i = 0; i < steps; i++) k = t[k];
    if (k == 0xFFFFFFFFu) sink[0] = k;
}
int main()
{
    uint32_t *d_ids, h_ids[64];
    hipMalloc(&d_ids, 256);
    k_ids<<<64, 64>>>(d_ids);
    hipMemcpy(h_ids, d_ids, 256, hipMemcpyDeviceToHost);
    printf("XCC_ID of workgroups 0..31:");
    for (int i = 0; i < 32; i++) printf(" %u", h_ids[i]);
    printf("\n");
    std::mt19937 rng(1);
    for (size_t mb10 : {5, 10, 20, 36, 50, 80, 160}) {
        const uint32_t entries = (uint32_t)(mb10 * 1024 * 1024 / 10 / 4);
        std::vector<uint32_t> perm(entries), tab((size_t)entries * 8);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int x = 0; x < 8; x++)
            for (uint32_t i = 0; i < entries; i++) tab[(size_t)x * entries + perm[i]] = perm[(i + 1) % entries];   // one big cycle
        uint32_t *d_tab, *d_sink;
        hipMalloc(&d_tab, tab.size() * 4);
        hipMalloc(&d_sink, 4);
        hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
        for (int use_xcc = 0; use_xcc < 2; use_xcc++)
            for (int wgs : {8, 32}) {
                const uint32_t steps = 2000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                k_walk<<<8 * wgs, 512>>>(d_tab, entries, steps, d_sink, use_xcc); hipDeviceSynchronize();
                hipEventRecord(e0); k_walk<<<8 * wgs, 512>>>(d_tab, entries, steps, d_sink, use_xcc); hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("table %4.1f MB per XCD, %2d WGs per XCD, %s: %7.3f ms, %6.1f ns per dependent step, %6.2f G reads/s\n", mb10 / 10.0, wgs,
                       use_xcc ? "XCC_ID     " : "blockIdx % 8", ms, ms * 1e6 / steps, 8.0 * wgs * 512 * steps / ms / 1e6);
            }
        hipFree(d_tab); hipFree(d_sink);
    }
    return 0;
}
