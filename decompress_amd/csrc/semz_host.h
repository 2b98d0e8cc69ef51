// semz_host.h -- host-side internals of the library (context, kernel launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include "semz_device.h"

struct semz_check_tables {
    uint32_t crc32[4][256];     // reflected 0xEDB88320, slice-by-4
    uint32_t bzcrc[4][256];     // MSB-first 0x04C11DB7, slice-by-4
    uint64_t crc64[4][256];     // reflected 0xC96C5795D7870F42, slice-by-4
    uint32_t x2n32[32];         // x^(2^k) mod P, reflected CRC-32 representation
    uint32_t x2nbz[32];         // same, MSB-first representation
    uint64_t x2n64[64];         // same, reflected CRC-64
};

struct semz_devbuf { void *p = nullptr; size_t cap = 0; };

struct semz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    semz_check_tables *d_tabs = nullptr;
    // this device's copy of the static scratch array of semz_decode_batch_device_noalloc (hipGetSymbolAddress)
    void *noalloc_scratch = nullptr;
    std::vector<semz_devbuf> bufs;      // grow-only scratch, indexed by SEMZ_SLOT_*
    // optional per-kernel timing (semz_set_option "profile"): event pairs on `stream`
    bool profile = false;
    // :window-size of the :raw-lzma2 format (lzma2.lisp:206-218), semz_set_option "window-size"
    uint64_t window_size = 0xFFFFFFFFull;
    // keywords of the :raw-lzma format (lzma.lisp:809-825): "lc", "lp", "pb", "decompressed-size" (< 0 = nil),
    // "eof-mode" (0 :maybe 1 :always 2 :never); "lzma-honour-size": see stream.cpp run_lzma
    int lz_lc = -1, lz_lp = -1, lz_pb = -1, lz_eof_mode = 0;
    long long lz_size = -1;
    bool lz_honour_size = false;
    uint32_t bz_overlap_blocks = 2048;  // option "bzip2-overlap-blocks": least blocks of a part of a large batch (0 = one piece)
    hipStream_t aux_stream = nullptr;   // semz_run_bzip2: the inverse BWT of one part of a large batch beside the decode of the next
    std::vector<hipEvent_t> aux_events;
    int bz_tables = 0;                  // option "bzip2-tables": 0 = by batch size, 1 = narrow root tables, 2 = wide (tests, measurements)
    bool lz_no_fast = false;            // option "lzma-no-fast": general kernel only (A/B measurements, tests of the general kernel)
    // semz_open_multi: contexts on the other devices; the stream calls spread a predicted batch over all of them
    std::vector<semz_ctx *> helpers;
    // Device buffers the stream calls have handed back (input copies, output arenas): a call that decodes a gigabyte
    // should not pay hipMalloc and hipFree for it every time.  At most SEMZ_POOL_MAX buffers, the smallest goes first.
    std::vector<semz_devbuf> pool;
    // The batch loop of the stream calls (stream.cpp: decode_windowed): two further contexts on this device, opened
    // on first use, so that one window of a large input uploads while another decodes and a third downloads.
    std::vector<semz_ctx *> lanes;
    // option "batch-budget": device bytes a predicted batch of a stream call may take (input, output, decoder
    // scratch); a larger batch is decoded window by window into host memory.  0 = three quarters of what the
    // device has free when the call starts.
    uint64_t batch_budget = 0;
    uint64_t last_windows = 0;          // windows the last stream call decoded (semz_get_option "windows")
    uint64_t last_uploaded = 0;         // bytes of input the last stream call sent to devices, all of them ("uploaded-bytes")
    uint64_t last_moved = 0;            // pieces of such windows the last read found out of place in the caller's buffer ("moved")
    // further streams for downloads of large results in parallel slices (created on first use)
    std::vector<hipStream_t> copy_streams;
    // Decoded streams that live on this context (semz_stream_decode): semz_close takes their device memory back and
    // cuts them loose, so that a release that comes after the close (a finalizer, a destructor) touches no context.
    // `mu` guards this list and the pool: a release may come from another thread than the one that decodes.
    std::vector<struct semz_decoded *> live;
    std::mutex mu;
    struct span { const char *name; hipEvent_t a, b; };
    std::vector<span> spans;
    char err[512] = {0};
};

enum {
    SEMZ_SLOT_TAILS = 0, SEMZ_SLOT_IN, SEMZ_SLOT_UNITS, SEMZ_SLOT_OUT, SEMZ_SLOT_RES,
    SEMZ_SLOT_BZ_META, SEMZ_SLOT_BZ_DATA, SEMZ_SLOT_BZ_TT, SEMZ_SLOT_BZ_AUX,
    SEMZ_SLOT_LZ_META, SEMZ_SLOT_LZ_WIDE, SEMZ_SLOT_STREAM_A, SEMZ_SLOT_STREAM_B, SEMZ_SLOT_STREAM_C,
    SEMZ_SLOT_COUNT
};

int semz_fail(semz_ctx *ctx, int rc, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
// semz_close: the decoded streams still alive on `c` give their device memory back and forget the context
void semz_stream_orphan_all(semz_ctx *c);
// RAII-less helpers: bracket a kernel (or kernel sequence) with events when profiling
void semz_span_begin(semz_ctx *c, const char *name);
void semz_span_end(semz_ctx *c);
int semz_scratch(semz_ctx *c, int slot, size_t bytes, void **p);
constexpr size_t SEMZ_POOL_MAX = 4;
// a device buffer of at least `bytes` (from the pool if one fits without wasting more than half); hipErrorOutOfMemory
// comes back as such after the pool has been emptied and the allocation tried again
hipError_t semz_pool_take(semz_ctx *c, size_t bytes, void **p, size_t *cap);
void semz_pool_give(semz_ctx *c, void *p, size_t cap);

extern "C" {
void semz_build_check_tables(semz_check_tables *t);
void semz_launch_inflate(const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                         uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails, hipStream_t stream);
void semz_launch_checksum(const uint8_t *d_out, const semz_unit *d_units, uint32_t nunits,
                          semz_result *d_results, const semz_tail *d_tails,
                          const semz_check_tables *d_tabs, uint32_t kind_mask, hipStream_t stream);
}
int semz_run_bzip2(semz_ctx *c, const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                   uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails);
// device scratch semz_run_bzip2 takes per block (the batch loop sizes its windows with it)
size_t semz_bzip2_scratch_per_block();
// the accept-only fast path for xz blocks / LZMA2 units (lzma2_fast.hip); what it leaves with the internal retry
// status is decoded by the general kernels of lzma2.hip
void semz_launch_lzma2_fast(const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits, uint8_t *d_out,
                            size_t out_cap, semz_result *d_results, semz_tail *d_tails, hipStream_t stream);
// units the fast kernel has handed to the general kernel on the current device since the library was loaded
int semz_lzma2_handed_back(long *value);
// wide: literal probabilities in HBM scratch (SEMZ_UNIT_LZMA_WIDE)
int semz_run_lzma2(semz_ctx *c, const uint8_t *d_in, size_t in_len, const semz_unit *d_units, uint32_t nunits,
                   uint8_t *d_out, size_t out_cap, semz_result *d_results, semz_tail *d_tails, bool wide);
