// semz_api.hip -- the C ABI of include/semz_gpu.h: context, device memory
// helpers, and the batched decode entry points that sequence the kernels.
//
// No CPU decode path exists in this library: every entry point that decodes
// needs a HIP device and fails (negative return + semz_last_error) without one.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "semz_host.h"

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return semz_fail(ctx, -2, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int semz_fail(semz_ctx *ctx, int rc, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return rc;
}

void semz_span_begin(semz_ctx *c, const char *name)
{
    if (!c->profile) return;
    semz_ctx::span s{name, nullptr, nullptr};
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    (void)hipEventRecord(s.a, c->stream);
    c->spans.push_back(s);
}
void semz_span_end(semz_ctx *c)
{
    if (!c->profile || c->spans.empty()) return;
    (void)hipEventRecord(c->spans.back().b, c->stream);
}

// Per-kernel timing of the calls made since the last reset (profile option on).
extern "C" int semz_profile_count(semz_ctx *c) { return c ? (int)c->spans.size() : 0; }
extern "C" int semz_profile_get(semz_ctx *c, int i, const char **name, float *ms)
{
    if (!c || i < 0 || (size_t)i >= c->spans.size()) return -1;
    auto &s = c->spans[i];
    if (hipEventSynchronize(s.b) != hipSuccess) return -2;
    if (hipEventElapsedTime(ms, s.a, s.b) != hipSuccess) return -2;
    if (name) *name = s.name;
    return 0;
}
extern "C" void semz_profile_reset(semz_ctx *c)
{
    if (!c) return;
    for (semz_ctx *h : c->helpers) semz_profile_reset(h);      // (spans recorded by the other devices and by the lanes
    for (semz_ctx *h : c->lanes) semz_profile_reset(h);        // of the batch loop while "profile" was on)
    for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    c->spans.clear();
}

// ---- scratch of the entry point that allocates nothing (semz_decode_batch_device_noalloc) ----
// The deflate-family kernels hand the trailer fields of every unit to the checksum kernel through a table of
// semz_tail records (64 bytes per unit), which semz_decode_batch_device keeps in grow-only hipMalloc'ed scratch.
// Callers that cannot allocate get it from this static array instead: 4 MiB per GPU the library runs on, held
// until the process ends.  In host code `scratch` is only a stand-in for the symbol; the array's address on a
// device comes from hipGetSymbolAddress (semz_open, with that device current) and lives in the context.
static __device__ __align__(256) unsigned char scratch[SEMZ_NOALLOC_MAX_UNITS * sizeof(semz_tail)];

extern "C" int semz_open(int device, semz_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -3;   // no GPU: nothing to fall back to
    if (device < 0 || device >= ndev) return -1;
    semz_ctx *c = new semz_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return -2; }
    // The second stream of semz_run_bzip2 (the inverse BWT of one part of a large batch beside the decode of the
    // next).  At the default priority and created here, not on first use: created at the highest priority once the
    // download streams of the stream calls existed (more streams than the 4 hardware queues of a process), the event
    // hand-over between the two streams never completed -- a 4800-block batch after one gzip stream call hung.  The
    // priority was worth nothing measurable (28.0 GB/s on that batch either way).
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) {
        hipStreamDestroy(c->stream);
        delete c;
        return -2;
    }
    semz_check_tables *h = new semz_check_tables();
    semz_build_check_tables(h);
    hipError_t e = hipMalloc((void **)&c->d_tabs, sizeof(semz_check_tables));
    if (e == hipSuccess) e = hipMemcpy(c->d_tabs, h, sizeof(semz_check_tables), hipMemcpyHostToDevice);
    delete h;
    if (e == hipSuccess) e = hipGetSymbolAddress(&c->noalloc_scratch, HIP_SYMBOL(scratch));   // this device's copy
    if (e != hipSuccess || !c->noalloc_scratch) {
        if (c->d_tabs) hipFree(c->d_tabs);
        hipStreamDestroy(c->aux_stream);
        hipStreamDestroy(c->stream);
        delete c;
        return -2;
    }
    *out = c;
    return 0;
}

extern "C" int semz_open_multi(const int *devices, int ndev, semz_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 255) return -1;
    semz_ctx *c = nullptr;
    int rc = semz_open(devices[0], &c);
    if (rc) return rc;
    for (int i = 1; i < ndev; i++) {
        semz_ctx *h = nullptr;
        rc = semz_open(devices[i], &h);
        if (rc) { semz_close(c); return rc; }
        c->helpers.push_back(h);
    }
    hipSetDevice(c->device);
    *out = c;
    return 0;
}

extern "C" int semz_device_count(const semz_ctx *c) { return c ? 1 + (int)c->helpers.size() : 0; }

extern "C" void semz_close(semz_ctx *c)
{
    if (!c) return;
    semz_stream_orphan_all(c);
    for (semz_ctx *h : c->helpers) semz_close(h);
    c->helpers.clear();
    for (semz_ctx *h : c->lanes) semz_close(h);
    c->lanes.clear();
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    semz_profile_reset(c);
    for (auto &b : c->bufs) if (b.p) hipFree(b.p);
    for (auto &b : c->pool) hipFree(b.p);
    for (auto s : c->copy_streams) hipStreamDestroy(s);
    for (auto e : c->aux_events) hipEventDestroy(e);
    if (c->aux_stream) hipStreamDestroy(c->aux_stream);
    if (c->d_tabs) hipFree(c->d_tabs);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *semz_last_error(const semz_ctx *c) { return c ? c->err : "no context"; }

extern "C" int semz_set_option(semz_ctx *c, const char *name, long value)
{
    if (!c || !name) return -1;
    for (semz_ctx *h : c->helpers) (void)semz_set_option(h, name, value);      // the other devices of semz_open_multi follow
    for (semz_ctx *h : c->lanes) (void)semz_set_option(h, name, value);        // and the lanes of the batch loop
    if (!strcmp(name, "profile")) { c->profile = value != 0; return 0; }
    if (!strcmp(name, "window-size")) {
        if (value < 0) return semz_fail(c, -1, "window-size must be an unsigned integer");   // check-type, lzma2.lisp:212
        c->window_size = (uint64_t)value;
        return 0;
    }
    // :raw-lzma keywords with the reference's check-type ranges (lzma.lisp:818-825)
    if (!strcmp(name, "lc")) { if (value < 0 || value > 8) return semz_fail(c, -1, "lc must be in 0..8"); c->lz_lc = (int)value; return 0; }
    if (!strcmp(name, "lp")) { if (value < 0 || value > 4) return semz_fail(c, -1, "lp must be in 0..4"); c->lz_lp = (int)value; return 0; }
    if (!strcmp(name, "pb")) { if (value < 0 || value > 4) return semz_fail(c, -1, "pb must be in 0..4"); c->lz_pb = (int)value; return 0; }
    if (!strcmp(name, "decompressed-size")) { c->lz_size = value < 0 ? -1 : value; return 0; }
    if (!strcmp(name, "eof-mode")) {
        if (value < 0 || value > 2) return semz_fail(c, -1, "eof-mode must be 0 (:maybe), 1 (:always) or 2 (:never)");
        c->lz_eof_mode = (int)value;
        return 0;
    }
    if (!strcmp(name, "lzma-honour-size")) { c->lz_honour_size = value != 0; return 0; }
    if (!strcmp(name, "lzma-no-fast")) { c->lz_no_fast = value != 0; return 0; }
    if (!strcmp(name, "bzip2-overlap-blocks")) {
        if (value < 0) return semz_fail(c, -1, "bzip2-overlap-blocks must not be negative");
        c->bz_overlap_blocks = (uint32_t)value;
        return 0;
    }
    if (!strcmp(name, "bzip2-tables")) {
        if (value < 0 || value > 2) return semz_fail(c, -1, "bzip2-tables must be 0 (by batch size), 1 (narrow) or 2 (wide)");
        c->bz_tables = (int)value;
        return 0;
    }
    if (!strcmp(name, "batch-budget")) {
        if (value < 0) return semz_fail(c, -1, "batch-budget must be a number of bytes (0 = automatic)");
        c->batch_budget = (uint64_t)value;
        return 0;
    }
    return semz_fail(c, -1, "unknown option %s", name);
}

extern "C" int semz_get_option(semz_ctx *c, const char *name, long *value)
{
    if (!c || !name || !value) return -1;
    if (!strcmp(name, "batch-budget")) { *value = (long)c->batch_budget; return 0; }
    if (!strcmp(name, "windows")) { *value = (long)c->last_windows; return 0; }
    if (!strcmp(name, "moved")) { *value = (long)c->last_moved; return 0; }
    if (!strcmp(name, "uploaded-bytes")) { *value = (long)c->last_uploaded; return 0; }
    if (!strcmp(name, "lzma-handed-back")) {
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess || semz_lzma2_handed_back(value))
            return semz_fail(c, -1, "cannot read the counter");
        return 0;
    }
    if (!strcmp(name, "profile")) { *value = c->profile; return 0; }
    if (!strcmp(name, "window-size")) { *value = (long)c->window_size; return 0; }
    return semz_fail(c, -1, "unknown option %s", name);
}

hipError_t semz_pool_take(semz_ctx *c, size_t bytes, void **p, size_t *cap)
{
    std::lock_guard<std::mutex> g(c->mu);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < c->pool.size(); i++)
        if (c->pool[i].cap >= bytes && c->pool[i].cap / 2 <= bytes && (best == SIZE_MAX || c->pool[i].cap < c->pool[best].cap)) best = i;
    if (best != SIZE_MAX) {
        *p = c->pool[best].p;
        *cap = c->pool[best].cap;
        c->pool.erase(c->pool.begin() + best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && !c->pool.empty()) {          // what the pool holds may be what is missing
        (void)hipGetLastError();
        for (auto &b : c->pool) (void)hipFree(b.p);
        c->pool.clear();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) *cap = bytes;
    return e;
}

void semz_pool_give(semz_ctx *c, void *p, size_t cap)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(c->mu);
    c->pool.push_back(semz_devbuf{p, cap});
    if (c->pool.size() > SEMZ_POOL_MAX) {
        size_t small = 0;
        for (size_t i = 1; i < c->pool.size(); i++)
            if (c->pool[i].cap < c->pool[small].cap) small = i;
        (void)hipFree(c->pool[small].p);
        c->pool.erase(c->pool.begin() + small);
    }
}

// grow-only device scratch owned by the context
int semz_scratch(semz_ctx *c, int slot, size_t bytes, void **p)
{
    if ((size_t)slot >= c->bufs.size()) c->bufs.resize(slot + 1);
    auto &b = c->bufs[slot];
    if (b.cap < bytes) {
        hipSetDevice(c->device);
        if (b.p) { hipStreamSynchronize(c->stream); hipFree(b.p); b.p = nullptr; b.cap = 0; }
        size_t cap = bytes + bytes / 4 + 256;
        HIPCHK(c, hipMalloc(&b.p, cap));
        b.cap = cap;
    }
    *p = b.p;
    return 0;
}

extern "C" int semz_decode_batch_device(semz_ctx *c, uint32_t kind, const void *d_in, size_t in_len,
                                        const semz_unit *d_units, size_t nunits, void *d_out,
                                        size_t out_cap, semz_result *d_results)
{
    if (!c) return -1;
    if (nunits == 0) return 0;
    if (!d_in || !d_units || !d_out || !d_results) return semz_fail(c, -1, "null device pointer");
    if (nunits > 0xFFFFFFF0u) return semz_fail(c, -1, "too many units");
    HIPCHK(c, hipSetDevice(c->device));
    void *tails = nullptr;
    int rc = semz_scratch(c, SEMZ_SLOT_TAILS, nunits * sizeof(semz_tail), &tails);
    if (rc) return rc;
    switch (kind) {
    case SEMZ_UNIT_DEFLATE:
    case SEMZ_UNIT_ZLIB:
    case SEMZ_UNIT_GZIP:
        semz_span_begin(c, "inflate_kernel");
        semz_launch_inflate((const uint8_t *)d_in, in_len, d_units, (uint32_t)nunits, (uint8_t *)d_out, out_cap,
                            d_results, (semz_tail *)tails, c->stream);
        semz_span_end(c);
        semz_span_begin(c, "checksum_kernel");
        semz_launch_checksum((const uint8_t *)d_out, d_units, (uint32_t)nunits, d_results, (const semz_tail *)tails,
                             c->d_tabs, (1u << SEMZ_TK_CRC32) | (1u << SEMZ_TK_ADLER32), c->stream);
        semz_span_end(c);
        break;
    case SEMZ_UNIT_BZIP2_BLOCK:
        rc = semz_run_bzip2(c, (const uint8_t *)d_in, in_len, d_units, (uint32_t)nunits, (uint8_t *)d_out, out_cap,
                            d_results, (semz_tail *)tails);
        if (rc) return rc;
        break;
    case SEMZ_UNIT_XZ_BLOCK:
    case SEMZ_UNIT_LZMA2:
    case SEMZ_UNIT_LZMA:
    case SEMZ_UNIT_LZMA_WIDE:
        rc = semz_run_lzma2(c, (const uint8_t *)d_in, in_len, d_units, (uint32_t)nunits, (uint8_t *)d_out, out_cap,
                            d_results, (semz_tail *)tails, kind == SEMZ_UNIT_LZMA_WIDE);
        if (rc) return rc;
        break;
    default:
        return semz_fail(c, -1, "unknown unit kind %u", kind);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Same as semz_decode_batch_device for the deflate family, without any allocation: no hipMalloc / hipFree, no
// event creation ("profile" spans are not recorded), the tail table lives in this GPU's static scratch array.
// Every call on one GPU shares that array: calls that overlap on one GPU (two contexts, two streams) race.
extern "C" int semz_decode_batch_device_noalloc(semz_ctx *c, uint32_t kind, const void *d_in, size_t in_len,
                                                const semz_unit *d_units, size_t nunits, void *d_out,
                                                size_t out_cap, semz_result *d_results)
{
    if (!c) return -1;
    if (nunits == 0) return 0;
    if (!d_in || !d_units || !d_out || !d_results) return semz_fail(c, -1, "null device pointer");
    if (kind > SEMZ_UNIT_GZIP)
        return semz_fail(c, -1, "semz_decode_batch_device_noalloc decodes deflate / zlib / gzip units only (kind %u needs "
                                "per-batch scratch: use semz_decode_batch_device)", kind);
    if (nunits > SEMZ_NOALLOC_MAX_UNITS)
        return semz_fail(c, -1, "semz_decode_batch_device_noalloc takes at most %zu units per call (got %zu): split the batch",
                         (size_t)SEMZ_NOALLOC_MAX_UNITS, nunits);
    HIPCHK(c, hipSetDevice(c->device));
    semz_tail *tails = (semz_tail *)c->noalloc_scratch;          // the device address, never the symbol's name
    // the array keeps the last call's contents: the records of this call start from zero
    HIPCHK(c, hipMemsetAsync(tails, 0, nunits * sizeof(semz_tail), c->stream));
    semz_launch_inflate((const uint8_t *)d_in, in_len, d_units, (uint32_t)nunits, (uint8_t *)d_out, out_cap,
                        d_results, tails, c->stream);
    semz_launch_checksum((const uint8_t *)d_out, d_units, (uint32_t)nunits, d_results, tails,
                         c->d_tabs, (1u << SEMZ_TK_CRC32) | (1u << SEMZ_TK_ADLER32), c->stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" int semz_sync(semz_ctx *c)
{
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" void *semz_stream(semz_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int semz_decode_batch(semz_ctx *c, const uint8_t *in, size_t in_len, const semz_unit *units,
                                 size_t nunits, uint8_t *out, size_t out_cap, semz_result *results)
{
    if (!c) return -1;
    if (nunits == 0) return 0;
    if (!in || !units || !results || (!out && out_cap)) return semz_fail(c, -1, "null pointer");
    // host-side validation of every span the kernels will touch
    uint64_t out_hi = 0;
    uint32_t batch_kind = units[0].kind;
    uint32_t fam = units[0].kind <= SEMZ_UNIT_GZIP ? 0 : units[0].kind == SEMZ_UNIT_BZIP2_BLOCK ? 1 : 2;
    for (size_t i = 0; i < nunits; i++) {
        const semz_unit &u = units[i];
        uint32_t f = u.kind <= SEMZ_UNIT_GZIP ? 0 : u.kind == SEMZ_UNIT_BZIP2_BLOCK ? 1 : 2;
        if (u.kind > SEMZ_UNIT_LZMA_WIDE || f != fam) return semz_fail(c, -1, "unit %zu: mixed or unknown kind", i);
        if (u.kind == SEMZ_UNIT_LZMA_WIDE) batch_kind = SEMZ_UNIT_LZMA_WIDE;
        if (u.in_off > u.in_limit || u.in_limit > in_len) return semz_fail(c, -1, "unit %zu: input span out of range", i);
        if (u.in_bit_off > 7) return semz_fail(c, -1, "unit %zu: bit offset", i);
        if (u.out_off == SEMZ_OUT_AUTO) {
            if (u.kind != SEMZ_UNIT_BZIP2_BLOCK) return semz_fail(c, -1, "unit %zu: SEMZ_OUT_AUTO is for bzip2 blocks", i);
            out_hi = out_cap;
            continue;
        }
        if (u.out_off > out_cap || u.out_cap > out_cap - u.out_off) return semz_fail(c, -1, "unit %zu: output span out of range", i);
        if (u.prefix_len && (u.prefix_off > in_len || u.prefix_len > in_len - u.prefix_off))
            return semz_fail(c, -1, "unit %zu: prefix span out of range", i);
        if (u.in_bit_off > 7) return semz_fail(c, -1, "unit %zu: bit offset", i);
        if (u.out_off + u.out_cap > out_hi) out_hi = u.out_off + u.out_cap;
    }
    HIPCHK(c, hipSetDevice(c->device));
    void *d_in, *d_units, *d_out, *d_res;
    int rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_IN, in_len + 16, &d_in))) return rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_UNITS, nunits * sizeof(semz_unit), &d_units))) return rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_OUT, out_hi + 16, &d_out))) return rc;
    if ((rc = semz_scratch(c, SEMZ_SLOT_RES, nunits * sizeof(semz_result), &d_res))) return rc;
    HIPCHK(c, hipMemcpyAsync(d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_units, units, nunits * sizeof(semz_unit), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_res, 0xFF, nunits * sizeof(semz_result), c->stream));
    rc = semz_decode_batch_device(c, batch_kind, d_in, in_len, (const semz_unit *)d_units, nunits, d_out, out_hi,
                                  (semz_result *)d_res);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, d_res, nunits * sizeof(semz_result), hipMemcpyDeviceToHost, c->stream));
    if (out_hi) HIPCHK(c, hipMemcpyAsync(out, d_out, out_hi, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int semz_dev_alloc(semz_ctx *c, size_t bytes, void **p)
{
    if (!c || !p) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, bytes ? bytes : 1));
    return 0;
}
extern "C" int semz_dev_free(semz_ctx *c, void *p)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(p));
    return 0;
}
extern "C" int semz_dev_upload(semz_ctx *c, void *d, const void *s, size_t n)
{
    if (!c) return -1;
    HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int semz_dev_download(semz_ctx *c, void *d, const void *s, size_t n)
{
    if (!c) return -1;
    HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int semz_event_create(semz_ctx *c, void **ev)
{
    if (!c || !ev) return -1;
    hipEvent_t e;
    HIPCHK(c, hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}
extern "C" int semz_event_record(semz_ctx *c, void *ev)
{
    if (!c) return -1;
    HIPCHK(c, hipEventRecord((hipEvent_t)ev, c->stream));
    return 0;
}
extern "C" int semz_event_elapsed_ms(semz_ctx *c, void *a, void *b, float *ms)
{
    if (!c || !ms) return -1;
    HIPCHK(c, hipEventSynchronize((hipEvent_t)b));
    HIPCHK(c, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}
extern "C" int semz_event_destroy(semz_ctx *c, void *ev)
{
    if (!c) return -1;
    HIPCHK(c, hipEventDestroy((hipEvent_t)ev));
    return 0;
}
extern "C" void semz_free(void *p) { free(p); }
