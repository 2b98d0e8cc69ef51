// Developer microbenchmark: what pinning a caller's pageable buffer for one call costs (hipHostRegister +
// hipHostUnregister) against what it buys (device-to-host copy rate into registered memory), for the sizes
// of the end-to-end calls in bench.py.  VERDICT r2 #3 asked for the measurement.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    for (size_t mb : {64, 256, 1024}) {
        const size_t n = mb << 20;
        void *d = nullptr;
        if (hipMalloc(&d, n) != hipSuccess) return 1;
        hipMemset(d, 1, n);
        char *h = (char *)aligned_alloc(2u << 20, n);
        memset(h, 2, n);                                     // touched, like a caller's buffer
        hipDeviceSynchronize();
        double t0 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        double t_page = now() - t0;
        t0 = now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
        double t_reg = now() - t0;
        if (e != hipSuccess) {
            printf("%zu MiB: hipHostRegister failed: %s", mb, hipGetErrorString(e));
            (void)hipGetLastError();
            e = hipHostRegister(h, n, hipHostRegisterPortable | hipHostRegisterMapped);
            printf("; with Portable | Mapped: %s; D2H pageable %.1f ms (%.1f GB/s)\n", hipGetErrorString(e), t_page * 1e3, n / t_page / 1e9);
            if (e != hipSuccess) { (void)hipGetLastError(); free(h); hipFree(d); continue; }
        }
        t0 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        double t_pin = now() - t0;
        t0 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice);
        double t_pin_up = now() - t0;
        t0 = now();
        hipHostUnregister(h);
        double t_unreg = now() - t0;
        t0 = now();
        hipMemcpy(d, h, n, hipMemcpyHostToDevice);
        double t_page_up = now() - t0;
        printf("%5zu MiB: D2H pageable %6.1f ms (%5.1f GB/s)  register %6.1f ms  D2H registered %6.1f ms (%5.1f GB/s)  "
               "H2D registered %6.1f ms  unregister %6.1f ms  H2D pageable %6.1f ms\n",
               mb, t_page * 1e3, n / t_page / 1e9, t_reg * 1e3, t_pin * 1e3, n / t_pin / 1e9, t_pin_up * 1e3, t_unreg * 1e3, t_page_up * 1e3);
        fflush(stdout);
        free(h);
        hipFree(d);
    }
    return 0;
}
