// stream.cpp -- semz_decode_stream: decompress / decompress-all for a fully
// buffered input (interface.lisp:149-208) on top of the batched unit decoder.
//
// Strategy for every format: find the units with the host splitter, decode all
// of them in one device batch, then walk the container forward on the host in
// exactly the order of the reference's `refill` state machine
// (interface.lisp:91-133) and of its per-format states (gzip.lisp:89-115,
// bzip2.lisp:456-503, xz.lisp:381-457, 166-233, 118-134), consuming the
// already decoded units.  The walk reproduces "first error wins": a unit's
// failure surfaces when the walk reaches it, outputs of earlier units stay
// valid.  A unit the splitter did not predict (no index, a lying index, a
// false magic) is decoded on demand as a batch of one.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "container.h"
#include "multi.h"
#include "semz_host.h"

using namespace semz;

namespace {

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return semz_fail(ctx, -2, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct Span { uint64_t off, len; };

struct Run {
    semz_ctx *c;
    const uint8_t *in;
    size_t in_len;
    uint8_t *d_in = nullptr;                   // (input copy and arena come from the context's pool and go back to it)
    size_t d_in_cap = 0;
    uint8_t *d_out = nullptr;
    size_t out_cap = 0, out_used = 0;          // device output arena; decoded spans are appended
    semz_unit *d_units = nullptr;
    semz_result *d_res = nullptr;
    size_t unit_cap = 0;
    std::vector<Span> spans;                   // output of the stream, in order, as arena spans
    uint64_t emitted = 0;                      // bytes in `spans`
    std::vector<semz_member> members;          // members the walk has completed, in order
    // The other devices of a context opened with semz_open_multi: each decodes a contiguous range of a predicted
    // batch from the bytes of that range and keeps the output in an arena of its own; offsets into it carry the
    // helper's index (1..) in their top byte (multi.h).  Everything the walk decodes on demand runs on the
    // primary device, which holds the whole input.
    struct Helper {
        semz_ctx *c = nullptr;
        uint8_t *d_in = nullptr, *d_out = nullptr;
        size_t in_cap = 0, out_cap = 0, out_used = 0;
        semz_unit *d_units = nullptr;
        semz_result *d_res = nullptr;
        size_t unit_cap = 0;
    };
    std::vector<Helper> helpers;
    // The primary device's own range of a batch that is spread over several devices: it is decoded from the bytes
    // of that range only, like the helpers' ranges (the whole input goes up only when the walk has to decode
    // something by itself, see ensure_input).
    uint8_t *d_part = nullptr;
    size_t d_part_cap = 0;
    std::atomic<uint64_t> uploaded{0};         // bytes of input this call has sent to devices, all of them (option "uploaded-bytes")
    // The batch loop (interface.lisp:91-133, common.lisp:255-257: bounded memory, nothing
    // delivered is lost before an error).  A predicted batch that does not fit `budget` bytes of device memory is
    // decoded window by window (multi.h: window_units): every window uploads its own bytes of the input, decodes
    // into a device arena of its own and goes down into host memory, on one of several lanes -- two contexts on
    // the primary device and the helper devices -- so that the upload of one window, the kernels of another and
    // the download of a third overlap.  Window w of a call lives in chunks[w]; results point there with device
    // tag WINDOW_TAG and offset w << 40 | offset in the chunk.  A chunk is a piece of the caller's buffer when the
    // call has one and the predicted layout is the stream's layout (semz_decode_stream_into: nothing is copied
    // twice), else host memory of the Run.  The walk afterwards is the same: it takes what it reaches, decodes
    // on demand what the prediction missed, and stops at the first error with everything before it delivered.
    static constexpr unsigned WINDOW_TAG = 255;
    static constexpr int WINDOW_SHIFT = 40;
    struct HostChunk { uint8_t *p = nullptr; uint64_t len = 0; bool owned = false; };
    std::vector<HostChunk> chunks;
    std::vector<Helper> lanes;
    uint8_t *sink = nullptr;                   // the caller's output buffer, if the call has one
    size_t sink_cap = 0;
    uint64_t budget = 0;                       // device bytes a predicted batch may take
    size_t windows_run = 0;                    // (trace)
    std::vector<uint8_t> distrust;             // per unit of the last decode(): the result must not be used (see multi.h)
    std::vector<uint64_t> cand;                // gzip: member-start candidates found while the input was on its way up
    bool cand_ready = false;
    std::vector<BzCandidate> bz_cand;          // bzip2: the magics, found while the input was on its way up

    ~Run()
    {
        free_device();
        for (auto &k : chunks)
            if (k.owned) free(k.p);
    }
    // Everything the Run holds on its devices goes back; the Run forgets its context (semz_close calls this for
    // the decoded streams that outlive their context: nothing may touch the context after that).
    void free_device()
    {
        if (!c) return;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        semz_pool_give(c, d_in, d_in_cap);
        semz_pool_give(c, d_part, d_part_cap);
        semz_pool_give(c, d_out, out_cap);
        if (d_units) (void)hipFree(d_units);
        if (d_res) (void)hipFree(d_res);
        d_in = d_out = d_part = nullptr; d_units = nullptr; d_res = nullptr;
        d_in_cap = out_cap = unit_cap = d_part_cap = 0;
        for (auto &h : helpers) {
            (void)hipSetDevice(h.c->device);
            (void)hipStreamSynchronize(h.c->stream);
            semz_pool_give(h.c, h.d_in, h.in_cap);
            semz_pool_give(h.c, h.d_out, h.out_cap);
            if (h.d_units) (void)hipFree(h.d_units);
            if (h.d_res) (void)hipFree(h.d_res);
        }
        release_lanes();
        if (!helpers.empty()) (void)hipSetDevice(c->device);
        helpers.clear();
        c = nullptr;
    }
    void release_lanes()
    {
        for (auto &h : lanes) {
            (void)hipSetDevice(h.c->device);
            (void)hipStreamSynchronize(h.c->stream);
            // (a lane of the primary device is a context of its own: its buffers go to the primary's pool, where
            // the next call looks for them)
            semz_ctx *pc = h.c->device == c->device ? c : h.c;
            pool_give_locked(pc, h.d_in, h.in_cap);
            pool_give_locked(pc, h.d_out, h.out_cap);
            if (h.d_units) (void)hipFree(h.d_units);
            if (h.d_res) (void)hipFree(h.d_res);
            h.d_in = h.d_out = nullptr;
            h.d_units = nullptr; h.d_res = nullptr;
            h.in_cap = h.out_cap = h.unit_cap = 0;
        }
        lanes.clear();
        (void)hipSetDevice(c->device);
    }
    int open()
    {
        HIPCHK(c, hipSetDevice(c->device));
        budget = c->batch_budget;
        if (!budget) {
            size_t fr = 0, tot = 0;
            HIPCHK(c, hipMemGetInfo(&fr, &tot));
            for (auto &b : c->pool) fr += b.cap;                  // (what the pool holds is the library's to use)
            budget = (uint64_t)fr / 4 * 3;
        }
        for (semz_ctx *hc : c->helpers) {
            Helper h;
            h.c = hc;
            helpers.push_back(h);
        }
        // an input that takes a good part of the budget by itself goes up window by window, or, if the format has
        // no units to predict, when the first decode asks for it; with several devices every device gets the bytes
        // of its own range, and the whole input goes to the primary device only if the walk needs it there
        if (in_len > budget / 4 || (!helpers.empty() && in_len >= (1u << 20))) return 0;
        return ensure_input();
    }
    int ensure_input()
    {
        if (d_in) return 0;
        hipError_t e = semz_pool_take(c, in_len + 64, (void **)&d_in, &d_in_cap);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); d_in = nullptr; return RC_OOM; }
        HIPCHK(c, e);
        HIPCHK(c, hipMemcpyAsync(d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
        uploaded += in_len;
        return 0;
    }
    // Sizes that come out of the data (declared sizes, index records, ISIZE fields) are untrusted: a request the
    // device cannot serve is RC_OOM, which the callers turn into "decode without the prediction" or into the
    // per-stream status SEMZ_E_NO_MEMORY, never into a failed call.
    static constexpr int RC_OOM = 1000;
    // room for `need` more bytes at the end of the arena; keeps what is there
    int reserve(size_t need)
    {
        if (need > (SIZE_MAX >> 2) || out_used > (SIZE_MAX >> 2)) return RC_OOM;      // the sums below cannot wrap
        if (out_used + need + 64 <= out_cap) return 0;
        // the first request of a call is usually the whole prediction: no slack then (and the pool's buffer of the
        // last call fits again); growth afterwards takes half as much again
        size_t ncap = out_used ? (out_used + need + 64) * 3 / 2 + (1 << 20) : need + 64;
        uint8_t *n = nullptr;
        hipError_t e = semz_pool_take(c, ncap, (void **)&n, &ncap);
        if (e == hipErrorOutOfMemory && out_used) {            // without the slack, then
            (void)hipGetLastError();
            ncap = out_used + need + 64;
            e = semz_pool_take(c, ncap, (void **)&n, &ncap);
        }
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RC_OOM; }
        HIPCHK(c, e);
        if (out_used) HIPCHK(c, hipMemcpyAsync(n, d_out, out_used, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        semz_pool_give(c, d_out, out_cap);
        d_out = n;
        out_cap = ncap;
        return 0;
    }
    // decodes units (out_off relative to the arena, or SEMZ_OUT_AUTO packed from arena_base) into res
    int decode(std::vector<semz_unit> &units, std::vector<semz_result> &res, uint64_t arena_base, uint64_t arena_room)
    {
        const size_t n = units.size();
        res.assign(n, semz_result{});
        distrust.assign(n, 0);
        if (!n) return 0;
        // every fixed span must lie inside the room this call hands to the kernels (they check it again per unit)
        if (arena_base > out_cap || arena_room > out_cap - arena_base) return semz_fail(c, -4, "internal: arena room beyond the arena");
        for (auto &u : units)
            if (u.out_off != SEMZ_OUT_AUTO && (u.out_off > arena_room || u.out_cap > arena_room - u.out_off))
                return semz_fail(c, -4, "internal: unit span beyond the arena room");
        // one input, several devices: contiguous ranges of the batch, one per device, side by side
        std::vector<MultiRange> ranges;
        if (!helpers.empty() && n >= 2 * (helpers.size() + 1) && in_len >= (1u << 20))
            ranges = partition_units(units.data(), n, in_len, (unsigned)helpers.size() + 1, 1u << 16, arena_room);
        if (ranges.size() < 2) return decode_on_primary(units.data(), n, res.data(), arena_base, arena_room);
        std::vector<int> rcs(ranges.size(), 0);
        std::vector<std::thread> th;
        for (size_t p = 1; p < ranges.size(); p++)
            th.emplace_back([&, p] { rcs[p] = decode_on_helper(helpers[p - 1], (unsigned)p, units.data(), ranges[p], arena_base, res.data()); });
        rcs[0] = decode_range_on_primary(units.data(), ranges[0], res.data(), arena_base, arena_room);
        for (auto &t : th) t.join();
        HIPCHK(c, hipSetDevice(c->device));
        if (getenv("SEMZ_TRACE"))
            for (size_t p = 0; p < ranges.size(); p++)
                fprintf(stderr, "multi: device %d takes units [%zu, %zu), uploads %llu bytes of %zu, output room %llu\n",
                        p ? helpers[p - 1].c->device : c->device, ranges[p].first, ranges[p].first + ranges[p].count,
                        (unsigned long long)(ranges[p].in_hi - ranges[p].in_lo), in_len, (unsigned long long)ranges[p].out_room);
        for (size_t p = 0; p < ranges.size(); p++) {
            if (rcs[p] == RC_OOM) {                 // a helper without the memory: its range is decoded by the walk, on demand
                for (size_t i = 0; i < ranges[p].count; i++) distrust[ranges[p].first + i] = 1;
            } else if (rcs[p]) {
                if (p) semz_fail(c, rcs[p], "device %d: %s", helpers[p - 1].c->device, helpers[p - 1].c->err);
                return rcs[p];
            }
        }
        return 0;
    }
    int decode_on_primary(const semz_unit *units, size_t n, semz_result *res, uint64_t arena_base, uint64_t arena_room)
    {
        {
            int rc = ensure_input();
            if (rc == RC_OOM) {                                   // (the input does not fit the device: nothing can be decoded from it here)
                for (size_t i = 0; i < n; i++) {
                    res[i] = semz_result{};
                    res[i].status = SEMZ_E_NO_MEMORY;
                    res[i].in_end_bit = 8 * units[i].in_off + units[i].in_bit_off;
                }
                return 0;
            }
            if (rc) return rc;
        }
        if (n > unit_cap) {
            if (d_units) HIPCHK(c, hipFree(d_units));
            if (d_res) HIPCHK(c, hipFree(d_res));
            d_units = nullptr; d_res = nullptr;
            unit_cap = n + n / 2 + 16;
            HIPCHK(c, hipMalloc((void **)&d_units, unit_cap * sizeof(semz_unit)));
            HIPCHK(c, hipMalloc((void **)&d_res, unit_cap * sizeof(semz_result)));
        }
        HIPCHK(c, hipMemcpyAsync(d_units, units, n * sizeof(semz_unit), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(d_res, 0xFF, n * sizeof(semz_result), c->stream));
        int rc = semz_decode_batch_device(c, units[0].kind, d_in, in_len, d_units, n, d_out + arena_base, arena_room, d_res);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(res, d_res, n * sizeof(semz_result), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    // The primary device's range of a batch that several devices share: the bytes of the range go up (not the whole
    // input), the units read them at local offsets and write the arena at the offsets the batch gave them.  A unit
    // that fails on a span the range cut short is marked for the walk (which decodes it where the whole input is).
    int decode_range_on_primary(const semz_unit *units, const MultiRange &r, semz_result *res, uint64_t arena_base, uint64_t arena_room)
    {
        if (d_in) return decode_on_primary(units, r.count, res, arena_base, arena_room);      // (the whole input is up already)
        const size_t span = (size_t)(r.in_hi - r.in_lo), n = r.count;
        if (span + 64 > d_part_cap) {
            semz_pool_give(c, d_part, d_part_cap);
            d_part = nullptr;
            d_part_cap = 0;
            hipError_t e = semz_pool_take(c, span + 64, (void **)&d_part, &d_part_cap);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); d_part = nullptr; return RC_OOM; }
            HIPCHK(c, e);
        }
        HIPCHK(c, hipMemcpyAsync(d_part, in + r.in_lo, span, hipMemcpyHostToDevice, c->stream));
        uploaded += span;
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(units, r, local, clamped);
        for (auto &x : local)
            if (x.out_off != SEMZ_OUT_AUTO) x.out_off += r.out_lo;         // (output offsets stay those of the batch)
        if (n > unit_cap) {
            if (d_units) HIPCHK(c, hipFree(d_units));
            if (d_res) HIPCHK(c, hipFree(d_res));
            d_units = nullptr; d_res = nullptr;
            unit_cap = n + n / 2 + 16;
            HIPCHK(c, hipMalloc((void **)&d_units, unit_cap * sizeof(semz_unit)));
            HIPCHK(c, hipMalloc((void **)&d_res, unit_cap * sizeof(semz_result)));
        }
        HIPCHK(c, hipMemcpyAsync(d_units, local.data(), n * sizeof(semz_unit), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(d_res, 0xFF, n * sizeof(semz_result), c->stream));
        int rc = semz_decode_batch_device(c, local[0].kind, d_part, span, d_units, n, d_out + arena_base, arena_room, d_res);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(res, d_res, n * sizeof(semz_result), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; i++) {
            res[i].in_end_bit += 8 * r.in_lo;
            distrust[r.first + i] = (uint8_t)(clamped[i] && res[i].status != SEMZ_OK);
        }
        return 0;
    }
    // One range of the batch on a helper device, in a thread of its own: upload of the range's bytes, the same
    // kernels, results back in the frame of the batch (multi.h).  The output stays in the helper's arena.
    int decode_on_helper(Helper &h, unsigned tag, const semz_unit *units, const MultiRange &r, uint64_t frame_base, semz_result *res)
    {
        semz_ctx *hc = h.c;
        HIPCHK(hc, hipSetDevice(hc->device));
        const size_t span = (size_t)(r.in_hi - r.in_lo);
        if (span + 64 > h.in_cap) {
            semz_pool_give(hc, h.d_in, h.in_cap);
            h.d_in = nullptr;
            h.in_cap = 0;
            hipError_t e = semz_pool_take(hc, span + 64, (void **)&h.d_in, &h.in_cap);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RC_OOM; }   // (the walk decodes this range on demand)
            HIPCHK(hc, e);
        }
        HIPCHK(hc, hipMemcpyAsync(h.d_in, in + r.in_lo, span, hipMemcpyHostToDevice, hc->stream));
        uploaded += span;
        if (r.out_room > (SIZE_MAX >> 2)) return RC_OOM;
        if (h.out_used + r.out_room + 64 > h.out_cap) {
            size_t ncap = h.out_used + (size_t)r.out_room + 64;
            uint8_t *nb = nullptr;
            hipError_t e = semz_pool_take(hc, ncap, (void **)&nb, &ncap);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RC_OOM; }
            HIPCHK(hc, e);
            if (h.out_used) HIPCHK(hc, hipMemcpyAsync(nb, h.d_out, h.out_used, hipMemcpyDeviceToDevice, hc->stream));
            HIPCHK(hc, hipStreamSynchronize(hc->stream));
            semz_pool_give(hc, h.d_out, h.out_cap);
            h.d_out = nb;
            h.out_cap = ncap;
        }
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(units, r, local, clamped);
        const size_t n = r.count;
        if (n > h.unit_cap) {
            if (h.d_units) HIPCHK(hc, hipFree(h.d_units));
            if (h.d_res) HIPCHK(hc, hipFree(h.d_res));
            h.d_units = nullptr; h.d_res = nullptr;
            h.unit_cap = n + n / 2 + 16;
            HIPCHK(hc, hipMalloc((void **)&h.d_units, h.unit_cap * sizeof(semz_unit)));
            HIPCHK(hc, hipMalloc((void **)&h.d_res, h.unit_cap * sizeof(semz_result)));
        }
        HIPCHK(hc, hipMemcpyAsync(h.d_units, local.data(), n * sizeof(semz_unit), hipMemcpyHostToDevice, hc->stream));
        HIPCHK(hc, hipMemsetAsync(h.d_res, 0xFF, n * sizeof(semz_result), hc->stream));
        int rc = semz_decode_batch_device(hc, local[0].kind, h.d_in, span, h.d_units, n, h.d_out + h.out_used, (size_t)r.out_room, h.d_res);
        if (rc) return rc;
        std::vector<semz_result> lres(n);
        HIPCHK(hc, hipMemcpyAsync(lres.data(), h.d_res, n * sizeof(semz_result), hipMemcpyDeviceToHost, hc->stream));
        HIPCHK(hc, hipStreamSynchronize(hc->stream));
        globalise_results(r, tag, h.out_used, frame_base, clamped, lres.data(), res, distrust.data());
        h.out_used += (size_t)r.out_room;
        return 0;
    }
    // ---- the batch loop ----
    // does a predicted batch of `total` output bytes and `scratch` bytes of decoder scratch go window by window?
    bool want_windows(uint64_t total, uint64_t scratch) const
    {
        const uint64_t need = (uint64_t)in_len + total + scratch;
        return need > budget || need < total;
    }
    // Decodes `units` window by window into host memory.  The results are in the frame of the batch like those of
    // decode(): the caller adds `frame_base` to every out_off and gets the tagged offset of the bytes.  No arena of
    // the primary device is used.  `out_need[i]`: the room unit i asks for (multi.h: window_units).
    int decode_windowed(std::vector<semz_unit> &units, std::vector<semz_result> &res, const std::vector<uint64_t> &out_need,
                        uint64_t frame_total, uint64_t frame_base, uint64_t unit_scratch)
    {
        const size_t n = units.size();
        res.assign(n, semz_result{});
        distrust.assign(n, 0);
        if (!n) return 0;
        if (lanes.empty()) {
            while (c->lanes.size() < 2) {
                semz_ctx *l = nullptr;
                if (semz_open(c->device, &l)) return semz_fail(c, -2, "batch loop: no further context on device %d", c->device);
                l->lz_no_fast = c->lz_no_fast;                        // (options set before the lanes existed)
                l->bz_tables = c->bz_tables;
                l->bz_overlap_blocks = c->bz_overlap_blocks;
                c->lanes.push_back(l);
            }
            for (semz_ctx *l : c->lanes) {
                Helper h;
                h.c = l;
                lanes.push_back(h);
            }
            for (auto &x : helpers) {                                 // (with buffers of the batch loop's own)
                Helper h;
                h.c = x.c;
                lanes.push_back(h);
            }
        }
        // two lanes share the primary device with the whole input, if that is resident
        const uint64_t resident = d_in ? std::min<uint64_t>(in_len, budget / 2) : 0;
        const uint64_t limit = std::max<uint64_t>((budget - resident) / 2, 1u << 20);
        const std::vector<MultiRange> wins = window_units(units.data(), n, in_len, out_need.data(), limit, 1u << 16, unit_scratch);
        if (wins.size() >= (1u << (MULTI_TAG_SHIFT - WINDOW_SHIFT))) return semz_fail(c, -4, "batch loop: %zu windows", wins.size());
        const size_t chunk0 = chunks.size();
        chunks.resize(chunk0 + wins.size());
        // the caller's buffer takes the windows where the stream will have them, if the batch is the stream so far
        const bool fixed = units[0].out_off != SEMZ_OUT_AUTO;
        const bool in_place = fixed && sink && emitted == 0 && chunk0 == 0 && frame_total <= sink_cap;
        std::atomic<size_t> next{0};
        std::vector<int> rcs(lanes.size(), 0);
        std::vector<uint8_t> oom(wins.size(), 0);
        auto work = [&](size_t li) {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= wins.size()) return;
                int rc = decode_window(lanes[li], units.data(), wins[k], chunk0 + k, in_place, frame_base, res.data());
                if (rc == RC_OOM) oom[k] = 1;
                else if (rc) {
                    rcs[li] = rc;
                    next.store(wins.size());
                    return;
                }
            }
        };
        std::vector<std::thread> th;
        for (size_t li = 1; li < lanes.size() && li < wins.size(); li++) th.emplace_back(work, li);
        work(0);
        for (auto &t : th) t.join();
        HIPCHK(c, hipSetDevice(c->device));
        windows_run += wins.size();
        for (size_t li = 0; li < lanes.size(); li++)
            if (rcs[li]) return semz_fail(c, rcs[li], "batch loop, device %d: %s", lanes[li].c->device, lanes[li].c->err);
        for (size_t k = 0; k < wins.size(); k++)
            if (oom[k])                                               // the walk decodes these on demand (or reports the memory)
                for (size_t i = 0; i < wins[k].count; i++) distrust[wins[k].first + i] = 1;
        return 0;
    }
    int decode_window(Helper &h, const semz_unit *units, const MultiRange &r, size_t chunk, bool in_place, uint64_t frame_base,
                      semz_result *res)
    {
        semz_ctx *hc = h.c;
        semz_ctx *pc = hc->device == c->device ? c : hc;              // whose pool the buffers come from
        HIPCHK(hc, hipSetDevice(hc->device));
        const size_t span = (size_t)(r.in_hi - r.in_lo);
        if (span + 64 > h.in_cap) {
            pool_give_locked(pc, h.d_in, h.in_cap);
            h.d_in = nullptr;
            h.in_cap = 0;
            hipError_t e = pool_take_locked(pc, span + 64, (void **)&h.d_in, &h.in_cap);
            if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RC_OOM; }
            HIPCHK(hc, e);
        }
        HIPCHK(hc, hipMemcpyAsync(h.d_in, in + r.in_lo, span, hipMemcpyHostToDevice, hc->stream));
        uploaded += span;
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(units, r, local, clamped);
        const size_t n = r.count;
        if (n > h.unit_cap) {
            if (h.d_units) HIPCHK(hc, hipFree(h.d_units));
            if (h.d_res) HIPCHK(hc, hipFree(h.d_res));
            h.d_units = nullptr; h.d_res = nullptr;
            h.unit_cap = n + n / 2 + 16;
            HIPCHK(hc, hipMalloc((void **)&h.d_units, h.unit_cap * sizeof(semz_unit)));
            HIPCHK(hc, hipMalloc((void **)&h.d_res, h.unit_cap * sizeof(semz_result)));
        }
        HIPCHK(hc, hipMemcpyAsync(h.d_units, local.data(), n * sizeof(semz_unit), hipMemcpyHostToDevice, hc->stream));
        std::vector<semz_result> lres(n);
        uint64_t room = r.out_room;
        for (int attempt = 0;; attempt++) {
            if (room > (SIZE_MAX >> 2)) return RC_OOM;
            if (room + 64 > h.out_cap) {
                pool_give_locked(pc, h.d_out, h.out_cap);
                h.d_out = nullptr;
                h.out_cap = 0;
                hipError_t e = pool_take_locked(pc, (size_t)room + 64, (void **)&h.d_out, &h.out_cap);
                if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return RC_OOM; }
                HIPCHK(hc, e);
            }
            HIPCHK(hc, hipMemsetAsync(h.d_res, 0xFF, n * sizeof(semz_result), hc->stream));
            int rc = semz_decode_batch_device(hc, local[0].kind, h.d_in, span, h.d_units, n, h.d_out, (size_t)room, h.d_res);
            if (rc) return rc;
            HIPCHK(hc, hipMemcpyAsync(lres.data(), h.d_res, n * sizeof(semz_result), hipMemcpyDeviceToHost, hc->stream));
            HIPCHK(hc, hipStreamSynchronize(hc->stream));
            // outputs the device packs (bzip2 blocks) may need more room than the estimate: RLE1 expands
            bool cap = false;
            if (!r.fixed)
                for (auto &x : lres) cap = cap || x.status == SEMZ_E_OUT_CAP;
            if (!cap || attempt == 3) break;
            room = sat4(room);
        }
        uint64_t used = 0;
        if (r.fixed) used = r.out_room;
        else
            for (auto &x : lres)
                if (x.status == SEMZ_OK && x.out_off + x.out_len > used) used = x.out_off + x.out_len;
        HostChunk &k = chunks[chunk];
        k.len = used;
        if (in_place) k.p = sink + r.out_lo;
        else if (used) {
            if (used >= (8u << 20)) {
                k.p = (uint8_t *)aligned_alloc(2u << 20, (used + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
                if (k.p) madvise(k.p, used, MADV_HUGEPAGE);
            } else k.p = (uint8_t *)malloc(used);
            if (!k.p) return semz_fail(hc, -5, "out of host memory (%llu bytes)", (unsigned long long)used);
            k.owned = true;
        }
        if (used) {
            int rc = download(hc, k.p, h.d_out, used);
            if (rc) return rc;
            HIPCHK(hc, hipStreamSynchronize(hc->stream));
        }
        globalise_results(r, WINDOW_TAG, (uint64_t)chunk << WINDOW_SHIFT, frame_base, clamped, lres.data(), res, distrust.data());
        return 0;
    }
    // A predicted batch of units with fixed spans laid out in [0, total): into the arena behind out_used, or,
    // beyond the budget, window by window.  RC_OOM: the arena could not be had.  The caller adds out_used to the
    // results' offsets either way and moves out_used on by `total` unless `windowed` comes back set.
    int predecode(std::vector<semz_unit> &units, std::vector<semz_result> &res, uint64_t total, bool &windowed)
    {
        windowed = want_windows(total, 0);
        if (windowed) {
            // (the sizes are claims of the data: host memory of the library's own is only spent on a prediction that
            // is plausible for the input; beyond that the walk decodes unit by unit)
            if (total > sink_cap && total > std::max<uint64_t>(64ull << 20, sat_mul64(64, in_len))) return RC_OOM;
            std::vector<uint64_t> need(units.size());
            for (size_t i = 0; i < units.size(); i++) need[i] = units[i].out_cap;
            return decode_windowed(units, res, need, total, out_used, 0);
        }
        int rc = reserve(total);
        if (rc) return rc;
        return decode(units, res, out_used, total);
    }
    static uint64_t sat_mul64(uint64_t a, uint64_t b) { uint64_t r; return __builtin_mul_overflow(a, b, &r) ? ~0ull : r; }
    static uint64_t sat4(uint64_t v) { return v > (~0ull >> 2) ? ~0ull : v * 4; }
    // (the lanes of the primary device take their buffers from one pool, from several threads: the pool locks itself)
    static hipError_t pool_take_locked(semz_ctx *pc, size_t bytes, void **p, size_t *cap) { return semz_pool_take(pc, bytes, p, cap); }
    static void pool_give_locked(semz_ctx *pc, void *p, size_t cap) { semz_pool_give(pc, p, cap); }
    // one unit with an unknown output size: retried with a larger span while it reports SEMZ_E_OUT_CAP.
    // `limit` is the largest span that can be of use (a declared size, or what the format can expand to from the
    // bytes that are left); a span the device cannot give is SEMZ_E_NO_MEMORY for the stream, not a failed call.
    int decode_one(semz_unit u, semz_result &r, uint64_t guess, uint64_t limit)
    {
        std::vector<semz_unit> us(1);
        std::vector<semz_result> rs;
        if (guess > limit) guess = limit;
        for (;;) {
            int rc = reserve(guess);
            if (rc == RC_OOM) {
                r = semz_result{};
                r.status = SEMZ_E_NO_MEMORY;
                r.in_end_bit = 8 * u.in_off + u.in_bit_off;
                return 0;
            }
            if (rc) return rc;
            u.out_off = u.out_off == SEMZ_OUT_AUTO ? SEMZ_OUT_AUTO : 0;
            u.out_cap = guess;
            us[0] = u;
            rc = decode(us, rs, out_used, guess);
            if (rc) return rc;
            r = rs[0];
            if (r.status != SEMZ_E_OUT_CAP || guess >= limit) break;
            guess = guess > limit / 4 ? limit : guess * 4;
        }
        r.out_off += out_used;                 // arena offset
        if (r.status == SEMZ_OK) out_used += r.out_len;
        return 0;
    }
    void member_done(uint64_t in_end) { members.push_back(semz_member{in_end, emitted}); }
    // what only the decode needs: the input copy and the unit tables (the arena stays for semz_stream_read)
    void drop_working_set()
    {
        (void)hipStreamSynchronize(c->stream);
        semz_pool_give(c, d_in, d_in_cap);
        semz_pool_give(c, d_part, d_part_cap);
        d_part = nullptr;
        d_in_cap = d_part_cap = 0;
        if (d_units) (void)hipFree(d_units);
        if (d_res) (void)hipFree(d_res);
        d_in = nullptr; d_units = nullptr; d_res = nullptr; unit_cap = 0;
        for (auto &h : helpers) {
            (void)hipSetDevice(h.c->device);
            (void)hipStreamSynchronize(h.c->stream);
            semz_pool_give(h.c, h.d_in, h.in_cap);
            if (h.d_units) (void)hipFree(h.d_units);
            if (h.d_res) (void)hipFree(h.d_res);
            h.d_in = nullptr; h.d_units = nullptr; h.d_res = nullptr; h.unit_cap = 0; h.in_cap = 0;
        }
        if (!helpers.empty()) (void)hipSetDevice(c->device);
        release_lanes();
        in = nullptr;                          // no host pointer is kept beyond the call
        sink = nullptr;                        // (chunks inside the caller's buffer are only read by the call itself)
    }
    // Device to host.  A copy into pageable memory is staged by the runtime on the calling thread and runs at the
    // speed of that thread's memcpy (17 GB/s here); a large one is therefore cut into slices that go side by side,
    // each on a thread and a stream of its own (the device that `x` belongs to is current).
    static int download(semz_ctx *x, uint8_t *dst, const uint8_t *src, uint64_t len)
    {
        constexpr uint64_t SLICE_MIN = 16u << 20;
        constexpr unsigned SLICES = 4;
        if (len < 2 * SLICE_MIN) {
            HIPCHK(x, hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToHost, x->stream));
            return 0;
        }
        HIPCHK(x, hipStreamSynchronize(x->stream));            // what is copied was written on this stream
        while (x->copy_streams.size() < SLICES) {
            hipStream_t st;
            HIPCHK(x, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            x->copy_streams.push_back(st);
        }
        const uint64_t per = ((len / SLICES) + 4095) & ~4095ull;
        hipError_t errs[SLICES];
        std::vector<std::thread> th;
        for (unsigned t = 0; t < SLICES; t++) {
            const uint64_t o = std::min<uint64_t>(len, per * t), n = std::min<uint64_t>(len - o, per);
            errs[t] = hipSuccess;
            if (!n) continue;
            th.emplace_back([=, &errs] {
                hipError_t e = hipSetDevice(x->device);
                if (e == hipSuccess) e = hipMemcpyAsync(dst + o, src + o, n, hipMemcpyDeviceToHost, x->copy_streams[t]);
                if (e == hipSuccess) e = hipStreamSynchronize(x->copy_streams[t]);
                errs[t] = e;
            });
        }
        for (auto &t : th) t.join();
        for (unsigned t = 0; t < SLICES; t++)
            if (errs[t] != hipSuccess) return semz_fail(x, -2, "download: %s", hipGetErrorString(errs[t]));
        return 0;
    }
    // bytes [off, off + len) of the decoded stream into dst (host memory)
    int read(uint64_t off, uint8_t *dst, uint64_t len)
    {
        if (off > emitted || len > emitted - off) return semz_fail(c, -1, "semz_stream_read: range beyond the decoded stream");
        // Windows of the batch loop are in host memory already, and when the call decodes into the caller's buffer
        // most of them are where they belong.  The ones that are not (the walk did not take the prediction as it
        // was) may lie where other spans have to go: they are set aside first.
        struct Moved { uint8_t *dst; std::vector<uint8_t> bytes; };
        std::vector<Moved> moved;
        {
            uint64_t at = 0, o = off, l = len;
            uint8_t *d = dst;
            for (auto &sp : spans) {
                if (!l) break;
                if (o < at + sp.len) {
                    const uint64_t skip = o - at, take = std::min<uint64_t>(l, sp.len - skip);
                    if (multi_dev(sp.off) == WINDOW_TAG) {
                        const HostChunk &k = chunks[multi_off(sp.off) >> WINDOW_SHIFT];
                        const uint8_t *src = k.p + (multi_off(sp.off) & ((1ull << WINDOW_SHIFT) - 1)) + skip;
                        if (!k.owned && src != d) moved.push_back(Moved{d, std::vector<uint8_t>(src, src + take)});
                    }
                    d += take; o += take; l -= take;
                }
                at += sp.len;
            }
        }
        // what comes from which device: the spans of one device go down one after the other, the devices side by
        // side, each on a host thread of its own (every device has its own link to the host)
        struct Copy { uint8_t *dst; const uint8_t *src; uint64_t len; };
        std::vector<std::vector<Copy>> per_dev(helpers.size() + 1);
        uint64_t at = 0;
        for (auto &sp : spans) {
            if (!len) break;
            if (off < at + sp.len) {
                const uint64_t skip = off - at, take = std::min<uint64_t>(len, sp.len - skip);
                const unsigned dev = multi_dev(sp.off);
                if (dev == WINDOW_TAG) {
                    const HostChunk &k = chunks[multi_off(sp.off) >> WINDOW_SHIFT];
                    if (k.owned) memcpy(dst, k.p + (multi_off(sp.off) & ((1ull << WINDOW_SHIFT) - 1)) + skip, take);
                } else if (dev == 0) per_dev[0].push_back(Copy{dst, d_out + sp.off + skip, take});
                else if (dev <= helpers.size()) per_dev[dev].push_back(Copy{dst, helpers[dev - 1].d_out + multi_off(sp.off) + skip, take});
                else return semz_fail(c, -4, "internal: span on device %u", dev);
                dst += take; off += take; len -= take;
            }
            at += sp.len;
        }
        std::vector<int> rcs(per_dev.size(), 0);
        auto run_dev = [&](size_t d) {
            semz_ctx *x = d ? helpers[d - 1].c : c;
            if (hipSetDevice(x->device) != hipSuccess) { rcs[d] = semz_fail(x, -2, "hipSetDevice failed"); return; }
            for (auto &k : per_dev[d]) {
                int rc = download(x, k.dst, k.src, k.len);
                if (rc) { rcs[d] = rc; return; }
            }
            if (hipStreamSynchronize(x->stream) != hipSuccess) rcs[d] = semz_fail(x, -2, "hipStreamSynchronize failed");
        };
        std::vector<std::thread> th;
        for (size_t d = 1; d < per_dev.size(); d++)
            if (!per_dev[d].empty()) th.emplace_back(run_dev, d);
        run_dev(0);
        for (auto &t : th) t.join();
        HIPCHK(c, hipSetDevice(c->device));
        for (size_t d = 0; d < per_dev.size(); d++)
            if (rcs[d]) return d ? semz_fail(c, rcs[d], "device %d: %s", helpers[d - 1].c->device, helpers[d - 1].c->err) : rcs[d];
        for (auto &m : moved) memcpy(m.dst, m.bytes.data(), m.bytes.size());
        c->last_moved = moved.size();
        return 0;
    }
    void emit(uint64_t off, uint64_t len)
    {
        if (!len) return;
        emitted += len;
        if (!spans.empty() && spans.back().off + spans.back().len == off) spans.back().len += len;
        else spans.push_back(Span{off, len});
    }
    int gather(uint8_t **out, size_t *out_len)
    {
        const size_t total = emitted;
        // large results: 2 MiB alignment and a huge-page hint, so that the first touch of a fresh gigabyte does not
        // cost a page fault per 4 KiB (the download is otherwise bound by them); still released with free()
        uint8_t *h;
        if (total >= (8u << 20)) {
            h = (uint8_t *)aligned_alloc(2u << 20, (total + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
            if (h) madvise(h, total, MADV_HUGEPAGE);
        } else h = (uint8_t *)malloc(total ? total : 1);
        if (!h) return semz_fail(c, -5, "out of host memory (%zu bytes)", total);
        int rc = read(0, h, total);
        if (rc) { free(h); return rc; }
        *out = h;
        *out_len = total;
        return 0;
    }
};

// What the stream's result says about its end.  `detail` is the first argument of the reference's message.  A
// message with a second argument finds it in `aux`, unless that argument is a checksum the decoder computed (the
// "got" of "wanted ..., got ..."), which is in `check` as in a unit's result (little-endian bytes, check_len of them).
struct Outcome {
    int status = SEMZ_OK;
    uint32_t detail = 0;
    uint64_t consumed = 0;
    uint32_t aux = 0;
    uint8_t check[32] = {0};
    uint32_t check_len = 0;
    void set_check32(uint32_t v) { for (int i = 0; i < 4; i++) check[i] = (uint8_t)(v >> (8 * i)); check_len = 4; }
};

// The failure of a unit becomes the failure of the stream.  Second arguments of messages that the kernels do not
// carry are worked out here from the unit's result and the input: deflate.lisp:330 (the reach the window allowed:
// bytes produced so far, preset history included, capped by the window), deflate.lisp:384 (the value NLEN should
// have had), gzip.lisp:77 (the header's real CRC-16), gzip.lisp:112 (the real size).
void fail_from_unit(Outcome &oc, const semz_result &r, const uint8_t *in, uint64_t in_len, uint64_t unit_start, uint64_t window,
                    uint64_t prefix_len)
{
    oc.status = r.status;
    oc.detail = r.detail;
    memcpy(oc.check, r.check, sizeof oc.check);
    oc.check_len = r.check_len;
    const uint64_t end = (r.in_end_bit + 7) >> 3;
    switch (r.status) {
    case SEMZ_E_DEFLATE_DISTANCE: oc.aux = (uint32_t)std::min<uint64_t>(r.out_len + prefix_len, window ? window : 32768); break;
    case SEMZ_E_DEFLATE_STORED_LEN:                               // LEN is the two bytes in front of NLEN, which was read last
        if (end >= 4 && end <= in_len) oc.aux = ((uint32_t)in[end - 4] | (uint32_t)in[end - 3] << 8) ^ 0xFFFFu;
        break;
    case SEMZ_E_GZIP_HEADER_CRC:                                  // the stored CRC-16 was read last
        if (end >= unit_start + 2 && end <= in_len) oc.aux = crc32_host(in + unit_start, end - 2 - unit_start) & 0xFFFFu;
        break;
    case SEMZ_E_GZIP_ISIZE: oc.aux = (uint32_t)r.out_len; break;
    default: break;
    }
}

inline uint64_t sat_mul(uint64_t a, uint64_t b) { uint64_t r; return __builtin_mul_overflow(a, b, &r) ? ~0ull : r; }
inline uint64_t sat_add(uint64_t a, uint64_t b) { uint64_t r; return __builtin_add_overflow(a, b, &r) ? ~0ull : r; }
// what `rem` bytes of input can expand to at most
inline uint64_t deflate_bound(uint64_t rem) { return sat_add(sat_mul(rem, 1032), 1024); }     // 258 bytes per two bits
inline uint64_t lzma_bound(uint64_t rem) { return sat_add(sat_mul(rem, 1ull << 18), 1 << 21); } // 2 MiB per 11-byte LZMA2 chunk

// Places that look like the start of a gzip member: 1F 8B 08, reserved flag bits clear, XFL one of the values
// deflate writers use (0, 2, 4) and OS a value RFC 1952 lists (0..13, 255).  They are hypotheses only (see
// run_deflate_family), so the test may be as strict as it likes: a member it misses is decoded by the walk,
// alone; a look-alike it lets through costs the member around it the same.  Ten bytes must match where three
// did, and look-alikes inside compressed data drop from a few per gigabyte to none in a thousand.  The scan is
// cut into pieces for four threads (memchr runs at the speed of one core's loads).
std::vector<uint64_t> scan_gzip_candidates(const uint8_t *in, uint64_t in_len)
{
    std::vector<uint64_t> all;
    if (in_len < 18) return all;
    const uint64_t last = in_len - 18;                            // last position a member can start at
    const unsigned parts = in_len >= (8u << 20) ? 4 : 1;
    std::vector<std::vector<uint64_t>> found(parts);
    auto scan = [&](unsigned t) {
        const uint64_t lo = (last + 1) / parts * t, hi = t + 1 == parts ? last + 1 : (last + 1) / parts * (t + 1);
        for (uint64_t p = lo; p < hi; p++) {
            const void *hit = memchr(in + p, 0x1F, hi - p);
            if (!hit) break;
            p = (uint64_t)((const uint8_t *)hit - in);
            if (in[p + 1] == 0x8B && in[p + 2] == 8 && in[p + 3] < 0x20 && (in[p + 8] == 0 || in[p + 8] == 2 || in[p + 8] == 4) &&
                (in[p + 9] <= 13 || in[p + 9] == 255))
                found[t].push_back(p);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < parts; t++) th.emplace_back(scan, t);
    scan(0);
    for (auto &t : th) t.join();
    for (auto &f : found) all.insert(all.end(), f.begin(), f.end());
    return all;
}

// ---- deflate / zlib / gzip ----
int run_deflate_family(Run &R, int fmt, bool all_members, Outcome &oc, semz_stream_info *info)
{
    const uint32_t kind = fmt == SEMZ_FMT_GZIP ? SEMZ_UNIT_GZIP : fmt == SEMZ_FMT_ZLIB ? SEMZ_UNIT_ZLIB : SEMZ_UNIT_DEFLATE;
    // predecode: BGZF-indexed gzip members
    std::map<uint64_t, semz_result> pre;
    if (fmt == SEMZ_FMT_GZIP && all_members) {
        semz_unit *su = nullptr;
        size_t sn = 0;
        semz_stream_info si;
        if (semz_split(fmt, R.in, R.in_len, nullptr, nullptr, 0, &su, &sn, &si) == 0 && si.reserved > 1) {
            // members whose size the BGZF index gives.  The spans come from the members' ISIZE fields, which are
            // only hypotheses: one that deflate cannot reach from the member's bytes is left to the walk below,
            // and so is a member that overflows its span (SEMZ_E_OUT_CAP is not a condition of the reference:
            // decoded alone the member gets the trailer error the reference reports).
            std::vector<semz_unit> units;
            uint64_t total = 0;
            for (size_t i = 0; i < si.reserved; i++) {
                semz_unit u = su[i];
                const uint64_t next = i + 1 < sn ? su[i + 1].in_off : R.in_len;
                if (u.out_cap > deflate_bound(next - u.in_off)) continue;
                u.out_off = total;
                total += u.out_cap;                               // back to back: the stream's output stays one span
                units.push_back(u);
            }
            std::vector<semz_result> res;
            bool windowed = false;
            int rc = units.size() > 1 ? R.predecode(units, res, total, windowed) : Run::RC_OOM;
            if (rc && rc != Run::RC_OOM) { free(su); return rc; }
            if (rc == 0) {
                for (size_t i = 0; i < units.size(); i++) {
                    if (res[i].status == SEMZ_E_OUT_CAP || R.distrust[i]) continue;
                    res[i].out_off += R.out_used;
                    pre[units[i].in_off] = res[i];
                }
                if (!windowed) R.out_used += total;
            }
        }
        free(su);
        if (pre.empty()) {
            // Plain concatenated members (pigz, cat of .gz files) carry no index: a member's end is only known
            // once it is decoded.  So every place that looks like a member header (1F 8B 08, reserved flag bits
            // clear) is taken as a candidate and all candidates are decoded in one batch, on the hypothesis that
            // the next candidate is the next member: then the four bytes in front of it are this member's ISIZE,
            // which sizes its output span.  The forward walk below (interface.lisp:91-133) only uses the result
            // of a candidate it actually reaches, and only if its span was large enough; everything else (a
            // look-alike inside compressed data, a member followed by one) is decoded again, alone.
            const bool trace = getenv("SEMZ_TRACE") != nullptr;
            auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const double tr0 = now();
            const uint8_t *in = R.in;
            std::vector<uint64_t> cand = R.cand_ready ? std::move(R.cand) : scan_gzip_candidates(in, R.in_len);
            if (cand.size() >= 2 && cand[0] == 0) {
                std::vector<semz_unit> units;
                uint64_t total = 0;
                for (size_t i = 0; i < cand.size(); i++) {
                    const uint64_t next = i + 1 < cand.size() ? cand[i + 1] : R.in_len;
                    const uint64_t span = next - cand[i];
                    if (span < 18) continue;
                    const uint64_t isize = (uint64_t)in[next - 4] | (uint64_t)in[next - 3] << 8 | (uint64_t)in[next - 2] << 16 |
                                           (uint64_t)in[next - 1] << 24;
                    if (isize > deflate_bound(span)) continue;       // deflate cannot expand further: not a member end
                    semz_unit u{};
                    u.kind = SEMZ_UNIT_GZIP;
                    u.in_off = cand[i];
                    u.in_limit = R.in_len;
                    u.out_off = total;
                    u.out_cap = isize;
                    total += (isize + 15) & ~15ull;
                    units.push_back(u);
                }
                // (look-alike headers in a large single member can hypothesise gigabytes each: speculation is only
                // worth a bounded amount of memory, otherwise the members are decoded one after the other)
                const double tr1 = now();
                if (R.want_windows(total, 0)) {
                    // (window by window into host memory: back to back without the padding, which is how the
                    // stream has the members if the hypotheses hold)
                    total = 0;
                    for (auto &u : units) {
                        u.out_off = total;
                        total += u.out_cap;
                    }
                }
                std::vector<semz_result> res;
                bool windowed = false;
                int rc = units.size() >= 2 && total <= std::max<uint64_t>(64ull << 20, 16 * R.in_len) ? R.predecode(units, res, total, windowed)
                                                                                                     : Run::RC_OOM;
                if (rc && rc != Run::RC_OOM) return rc;
                if (rc == 0) {
                    if (trace) fprintf(stderr, "gzip members: candidates %.1f ms (%zu), arena + upload wait + decode %.1f ms, %zu windows\n", tr1 - tr0, cand.size(), now() - tr1, R.windows_run);
                    for (size_t i = 0; i < units.size(); i++) {
                        if (res[i].status == SEMZ_E_OUT_CAP || R.distrust[i]) continue;        // the hypothesis did not hold
                        res[i].out_off += R.out_used;
                        pre[units[i].in_off] = res[i];
                    }
                    if (!windowed) R.out_used += total;
                }
            }
        }
    }
    const double tw0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    struct WalkTrace {
        double t0;
        ~WalkTrace() { if (getenv("SEMZ_TRACE")) fprintf(stderr, "gzip members: walk %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0); }
    } walk_trace{tw0};
    uint64_t pos = 0;
    for (uint32_t member = 0;; member++) {
        semz_result r;
        auto it = pre.find(pos);
        if (it != pre.end()) r = it->second;
        else {
            semz_unit u{};
            u.kind = kind;
            u.in_off = pos;
            u.in_limit = R.in_len;
            uint64_t guess = (R.in_len - pos) * 6 + 65536;
            int rc = R.decode_one(u, r, guess, deflate_bound(R.in_len - pos));
            if (rc) return rc;
        }
        if (member == 0 && info) {
            if (fmt == SEMZ_FMT_ZLIB) info->first_header[3] = r.aux;
            info->nmembers = 0;
        }
        oc.consumed = pos + ((r.in_end_bit - 8 * pos + 7) >> 3);
        if (r.status != SEMZ_OK) {
            // (zlib enforces the window its header declares, zlib.lisp:36, 53)
            const uint64_t window = fmt == SEMZ_FMT_ZLIB && pos < R.in_len && (R.in[pos] >> 4) <= 7 ? 256ull << (R.in[pos] >> 4) : 32768;
            fail_from_unit(oc, r, R.in, R.in_len, pos, window, 0);
            return 0;
        }
        R.emit(r.out_off, r.out_len);
        R.member_done(oc.consumed);
        if (info) info->nmembers = member + 1;
        pos = oc.consumed;
        if (!all_members || pos == R.in_len) return 0;            // buffer-stream-check-eof, interface.lisp:110-112
        if (fmt != SEMZ_FMT_GZIP) { oc.status = SEMZ_E_TRAILING_DATA; return 0; }   // interface.lisp:125-130
    }
}

// ---- bzip2 ----
int run_bzip2(Run &R, bool all_members, Outcome &oc, semz_stream_info *info)
{
    std::vector<BzCandidate> cand = R.cand_ready ? std::move(R.bz_cand) : scan_bzip2(R.in, R.in_len);
    std::vector<semz_unit> units;
    std::vector<uint64_t> start_bits;
    uint64_t guess = 0;
    for (auto &x : cand) {
        if (x.footer) continue;
        semz_unit u{};
        u.kind = SEMZ_UNIT_BZIP2_BLOCK;
        u.in_off = x.bit >> 3;
        u.in_bit_off = (uint32_t)(x.bit & 7);
        u.in_limit = R.in_len;
        u.out_off = SEMZ_OUT_AUTO;
        u.param0 = x.level;
        units.push_back(u);
        start_bits.push_back(x.bit);
        guess += (uint64_t)x.level * 100000 + 4096;
    }
    std::map<uint64_t, size_t> by_bit;
    std::vector<semz_result> res;
    uint64_t arena0 = R.out_used;
    const bool windowed = !units.empty() && R.want_windows(guess, sat_mul(units.size(), semz_bzip2_scratch_per_block()));
    if (windowed) {
        std::vector<uint64_t> need(units.size());
        for (size_t i = 0; i < units.size(); i++) need[i] = (uint64_t)units[i].param0 * 100000 + 4096;
        int rc = R.decode_windowed(units, res, need, guess, arena0, semz_bzip2_scratch_per_block());
        if (rc) return rc;
    }
    while (!windowed) {
        int rc = R.reserve(guess);
        if (rc == Run::RC_OOM) { oc.status = SEMZ_E_NO_MEMORY; return 0; }
        if (rc) return rc;
        rc = R.decode(units, res, arena0, guess);
        if (rc) return rc;
        bool cap = false;
        for (auto &r : res) cap = cap || r.status == SEMZ_E_OUT_CAP;
        if (!cap) break;
        guess = sat_mul(guess, 4);                                // RLE1 can expand far beyond the block size
    }
    for (size_t i = 0; i < units.size(); i++) {
        if (R.distrust[i] || res[i].status == SEMZ_E_OUT_CAP) continue;   // (decoded from a cut span: see multi.h; or a window's room ran out)
        by_bit[start_bits[i]] = i;
        const uint64_t end = arena0 + res[i].out_off + res[i].out_len;
        if (res[i].status == SEMZ_OK && multi_dev(arena0 + res[i].out_off) == 0 && end > R.out_used) R.out_used = end;
    }
    // forward walk (bzip2.lisp:425-503)
    uint64_t pos = 0;
    for (uint32_t member = 0;; member++) {
        uint32_t level = 0, det = 0;
        int st = parse_bzip2_header(R.in, pos, R.in_len, &level, &det);
        if (member == 0 && info) { info->first_header[0] = level; info->status = st; info->detail = det; }
        if (st != SEMZ_OK) {
            oc.status = st; oc.detail = det;
            oc.consumed = st == SEMZ_EOF ? R.in_len : pos + (st == SEMZ_E_BZ_MAGIC ? 2 : 4);
            return 0;
        }
        uint64_t bit = 8 * (pos + 4);
        uint32_t stream_crc = 0;
        for (;;) {
            if (8 * R.in_len - bit < 48) { oc.status = SEMZ_EOF; oc.consumed = R.in_len; return 0; }
            const uint64_t magic = read_bits_be(R.in, R.in_len, bit, 48);
            if (magic == BZ_BLOCK_MAGIC) {
                semz_result r;
                auto it = by_bit.find(bit);
                if (it != by_bit.end() && units[it->second].param0 == level) {
                    r = res[it->second];
                    r.out_off += arena0;
                } else {
                    semz_unit u{};
                    u.kind = SEMZ_UNIT_BZIP2_BLOCK;
                    u.in_off = bit >> 3;
                    u.in_bit_off = (uint32_t)(bit & 7);
                    u.in_limit = R.in_len;
                    u.out_off = SEMZ_OUT_AUTO;
                    u.param0 = level;
                    int rc = R.decode_one(u, r, (uint64_t)level * 100000 * 2, (uint64_t)level * 100000 * 52 + 1024);   // RLE1: 5 bytes -> 259
                    if (rc) return rc;
                }
                if (r.status != SEMZ_OK) {
                    fail_from_unit(oc, r, R.in, R.in_len, bit >> 3, 0, 0);
                    oc.consumed = (r.in_end_bit + 7) >> 3;
                    return 0;
                }
                R.emit(r.out_off, r.out_len);
                const uint32_t crc = (uint32_t)r.check[0] | ((uint32_t)r.check[1] << 8) | ((uint32_t)r.check[2] << 16) | ((uint32_t)r.check[3] << 24);
                stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;      // bzip2.lisp:44-47
                bit = r.in_end_bit;
            } else if (magic == BZ_FOOTER_MAGIC) {
                bit += 48;
                if (8 * R.in_len - bit < 32) { oc.status = SEMZ_EOF; oc.consumed = R.in_len; return 0; }
                const uint32_t want = (uint32_t)read_bits_be(R.in, R.in_len, bit, 32);
                bit += 32;
                if (want != stream_crc) {
                    oc.status = SEMZ_E_BZ_STREAM_CRC; oc.detail = want; oc.consumed = (bit + 7) >> 3;
                    oc.set_check32(stream_crc);
                    return 0;
                }
                break;
            } else {
                oc.status = SEMZ_E_BZ_BLOCK_MAGIC; oc.consumed = (bit + 48 + 7) >> 3;
                return 0;
            }
        }
        pos = (bit + 7) >> 3;                                     // flush-byte, bzip2.lisp:497
        oc.consumed = pos;
        R.member_done(pos);
        if (info) info->nmembers = member + 1;
        if (!all_members || pos == R.in_len) return 0;
    }
}

// ---- xz ----
int run_xz(Run &R, bool all_members, Outcome &oc, semz_stream_info *info)
{
    // predecode everything the index promises
    std::map<uint64_t, std::pair<semz_unit, semz_result>> pre;   // keyed by the offset of the LZMA2 data
    {
        semz_unit *su = nullptr;
        size_t sn = 0;
        semz_stream_info si;
        if (semz_split(SEMZ_FMT_XZ, R.in, R.in_len, nullptr, nullptr, 0, &su, &sn, &si) == 0 && sn) {
            // The index is a prediction, not a fact: semz_split bounds every record by what LZMA2 can expand to
            // and the sum cannot wrap; an arena the device cannot give means "no prediction" (the walk below then
            // decodes block by block and reports what the reference would, an index mismatch at the end).
            std::vector<semz_unit> units(su, su + sn);
            std::vector<semz_result> res;
            bool windowed = false;
            int rc = R.predecode(units, res, si.total_out, windowed);
            if (rc && rc != Run::RC_OOM) { free(su); return rc; }
            if (rc == 0) {
                for (size_t i = 0; i < sn; i++) {
                    if (R.distrust[i]) continue;
                    res[i].out_off += R.out_used;
                    pre[units[i].in_off] = std::make_pair(units[i], res[i]);
                }
                if (!windowed) R.out_used += si.total_out;
            }
        }
        free(su);
    }
    const uint8_t *in = R.in;
    const uint64_t n = R.in_len;
    uint64_t pos = 0;
    bool skip_magic = false;
    auto fail = [&](int st, uint32_t d, uint64_t at) { oc.status = st; oc.detail = d; oc.consumed = at; return 0; };
    for (uint32_t member = 0;; member++) {
        uint32_t ct = 0, det = 0;
        uint64_t hend = 0;
        int st = parse_xz_stream_header(in, pos, n, skip_magic, &ct, &hend, &det);
        if (member == 0 && info) { info->first_header[0] = ct; info->status = st; info->detail = det; }
        if (st != SEMZ_OK) return fail(st, det, hend);
        pos = hend;
        const uint32_t stream_flags = (uint32_t)ct << 8;
        std::vector<std::pair<uint64_t, uint64_t>> seen;          // (unpadded size, uncompressed size)
        for (;;) {                                                // :block/index, xz.lisp:389-407
            if (pos >= n) return fail(SEMZ_EOF, 0, n);
            if (in[pos] == 0) { pos++; break; }
            XzBlockHeader h = parse_xz_block_header(in, pos, n);
            if (h.status != SEMZ_OK) {
                if (h.status == SEMZ_E_XZ_BLOCK_HDR_CRC) oc.set_check32(h.detail2);
                else oc.aux = h.detail2;
                return fail(h.status, h.detail, h.status == SEMZ_EOF ? n : pos + h.header_size + 4);
            }
            const uint64_t data = pos + h.header_size + 4;
            semz_unit u{};
            u.kind = SEMZ_UNIT_XZ_BLOCK;
            u.in_off = data;
            u.in_limit = n;
            u.flags = (h.have_csize ? SEMZ_UF_HAVE_IN_SIZE : 0) | (h.have_usize ? SEMZ_UF_HAVE_OUT_SIZE : 0);
            u.size_in = h.csize; u.size_out = h.usize;
            u.param0 = h.dict_size; u.param1 = ct; u.param2 = h.bcj_offset; u.param3 = h.delta_dists; u.filters = h.filters;
            semz_result r;
            auto it = pre.find(data);
            bool reuse = false;
            if (it != pre.end()) {
                const semz_unit &q = it->second.first;
                reuse = q.flags == u.flags && q.size_in == u.size_in && q.size_out == u.size_out && q.param0 == u.param0 &&
                        q.param1 == u.param1 && q.param2 == u.param2 && q.param3 == u.param3 && q.filters == u.filters &&
                        it->second.second.status != SEMZ_E_OUT_CAP;
            }
            if (reuse) r = it->second.second;
            else {
                // a declared size is a hint for the span, never more than LZMA2 can make of the bytes that are left;
                // the first span is modest and grows while the block asks for more
                const uint64_t bound = lzma_bound(n - data);
                const uint64_t first = sat_add(sat_mul(n - data, 64), 1 << 20);
                uint64_t limit = h.have_usize && h.usize < bound ? h.usize : bound;
                int rc = R.decode_one(u, r, std::min(first, limit), limit);
                if (rc) return rc;
                if (r.status == SEMZ_E_OUT_CAP && limit < bound) {
                    // the declared size was too small for what the block holds: decode on to find the
                    // error the reference reports (a size mismatch at block end, or an earlier one)
                    rc = R.decode_one(u, r, first, bound);
                    if (rc) return rc;
                }
            }
            if (r.status != SEMZ_OK) {
                fail_from_unit(oc, r, in, n, data, 0, 0);
                return fail(r.status, r.detail, (r.in_end_bit + 7) >> 3);
            }
            R.emit(r.out_off, r.out_len);
            const uint64_t real_csize = ((uint64_t)r.detail << 32) | r.aux;
            seen.emplace_back(h.header_size + 4 + real_csize + xz_check_size(ct), r.out_len);
            pos = r.in_end_bit >> 3;
        }
        // finish-xz-member: index and footer (xz.lisp:166-233)
        {
            const uint64_t idx0 = pos - 1;
            std::vector<uint8_t> ridx;
            ridx.push_back(0);
            auto push = [&](uint64_t v) {
                for (;;) {
                    ridx.push_back((uint8_t)(v & 0x7F));
                    v >>= 7;
                    if (!v) return;
                    ridx.back() |= 0x80;
                }
            };
            uint64_t count, a, b;
            auto vfail = [&](int s) { return fail(s, 0, s == SEMZ_EOF ? n : pos); };
            if ((st = xz_varint(in, &pos, n, &count)) != SEMZ_OK) return vfail(st);
            push(count);
            bool mism = false;
            for (uint64_t k = 0; k < count; k++) {
                if ((st = xz_varint(in, &pos, n, &a)) != SEMZ_OK) return vfail(st);
                if ((st = xz_varint(in, &pos, n, &b)) != SEMZ_OK) return vfail(st);
                push(a);
                push(b);
                if (k < seen.size() && (a != seen[k].first || b != seen[k].second)) mism = true;
            }
            while (ridx.size() & 3) {
                if (pos >= n) return fail(SEMZ_EOF, 0, n);
                if (in[pos++] != 0) return fail(SEMZ_E_XZ_INDEX_PADDING, 0, pos);
                ridx.push_back(0);
            }
            if (count != seen.size()) return fail(SEMZ_E_XZ_INDEX_COUNT, 0, pos);
            if (mism) return fail(SEMZ_E_XZ_INDEX_SIZES, 0, pos);
            if (n - pos < 4) return fail(SEMZ_EOF, 0, n);
            const uint32_t icrc = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
            pos += 4;
            if (icrc != crc32_host(ridx.data(), ridx.size())) {
                oc.set_check32(crc32_host(ridx.data(), ridx.size()));
                return fail(SEMZ_E_XZ_INDEX_CRC, icrc, pos);
            }
            if (n - pos < 12) return fail(SEMZ_EOF, 0, n);
            const uint8_t *f = in + pos;
            pos += 12;
            const uint32_t fcrc = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
            const uint32_t bsz = (uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24);
            if (fcrc != crc32_host(f + 4, 6)) {
                oc.set_check32(crc32_host(f + 4, 6));
                return fail(SEMZ_E_XZ_FOOTER_CRC, fcrc, pos);
            }
            if (4ull * ((uint64_t)bsz + 1) != 4 + ridx.size()) return fail(SEMZ_E_XZ_BACKWARD_SIZE, 0, pos);
            if (stream_flags != ((uint32_t)f[8] | ((uint32_t)f[9] << 8))) return fail(SEMZ_E_XZ_FOOTER_FLAGS, 0, pos);
            if (f[10] != 0x59 || f[11] != 0x5A) return fail(SEMZ_E_XZ_FOOTER_MAGIC, 0, pos);
            (void)idx0;
        }
        oc.consumed = pos;
        R.member_done(pos);
        if (info) info->nmembers = member + 1;
        if (!all_members || pos == n) return 0;
        // make-reset-state: stream padding, then the next member (xz.lisp:118-134)
        for (;;) {
            const uint64_t left = n - pos;
            if (left == 0) { oc.consumed = n; return 0; }
            if (left < 4) return fail(SEMZ_E_XZ_PADDING, 0, pos);
            const uint8_t *b = in + pos;
            pos += 4;
            if (b[0] | b[1] | b[2] | b[3]) {
                if (b[0] == 0xFD && b[1] == 0x37 && b[2] == 0x7A && b[3] == 0x58) {
                    if (pos >= n) return fail(SEMZ_EOF, 0, n);
                    if (in[pos++] == 0x5A) {
                        if (pos >= n) return fail(SEMZ_EOF, 0, n);
                        if (in[pos++] == 0x00) { skip_magic = true; break; }
                    }
                }
                return fail(SEMZ_E_XZ_TRAILING_GARBAGE, 0, pos);
            }
        }
    }
}

// ---- :raw-lzma2 / :lzma2 (lzma2.lisp:205-227): one unit, no members ----
int run_lzma2(Run &R, int fmt, bool all_members, Outcome &oc, semz_stream_info *info)
{
    uint64_t start = 0;
    uint32_t dict = (uint32_t)(R.c->window_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : R.c->window_size);   // min(window-size, 2^32 - 1)
    if (fmt == SEMZ_FMT_LZMA2) {
        if (R.in_len < 1) { oc.status = SEMZ_EOF; return 0; }
        const uint32_t o = R.in[0];                        // parse-lzma2-dict-size, lzma2.lisp:40-46
        if (o > 40) { oc.status = SEMZ_E_LZMA2_DICT_BYTE; oc.detail = o; oc.consumed = 1; return 0; }
        dict = o == 40 ? 0xFFFFFFFFu : (2u | (o & 1)) << (11 + o / 2);
        start = 1;
    }
    if (info) info->first_header[0] = dict;                // header plist (:window-size n)
    semz_unit u{};
    u.in_off = start;
    u.in_limit = R.in_len;
    u.kind = SEMZ_UNIT_LZMA2;
    u.param0 = dict;
    semz_result r{};
    int rc = R.decode_one(u, r, std::max<uint64_t>(1 << 20, sat_mul(8, R.in_len)), lzma_bound(R.in_len));
    if (rc) return rc;
    oc.consumed = (r.in_end_bit + 7) / 8;
    if (r.status != SEMZ_OK) { oc.status = r.status; oc.detail = r.detail; return 0; }
    R.emit(r.out_off, r.out_len);
    R.member_done(oc.consumed);
    if (info) info->nmembers = 1;
    if (all_members && oc.consumed != R.in_len) oc.status = SEMZ_E_TRAILING_DATA;       // interface.lisp:125-130
    return 0;
}

// ---- :raw-lzma / :lzma (lzma.lisp:809-865): one unit, no members ----
// As written, the :lzma constructor hands the header's size on as :expected-remaining-data, a keyword the
// :raw-lzma method does not have (its name is :decompressed-size, lzma.lisp:811 vs 857): the size reaches the
// header plist only and the data must end with a marker.  That is the default here; the option
// "lzma-honour-size" decodes what the code evidently means (declared size, eof-mode :maybe).
int run_lzma(Run &R, int fmt, bool all_members, Outcome &oc, semz_stream_info *info)
{
    semz_unit u{};
    u.in_limit = R.in_len;
    uint64_t guess = std::max<uint64_t>(1 << 20, sat_mul(8, R.in_len));
    if (fmt == SEMZ_FMT_LZMA) {
        if (R.in_len < 1) { oc.status = SEMZ_EOF; return 0; }
        const uint32_t octet = R.in[0];                      // parse-lzma-props, lzma.lisp:837-843
        const uint32_t lc = octet % 9, lp = (octet / 9) % 5, pb = octet / 45;
        if (pb > 4) { oc.status = SEMZ_E_LZMA_PB; oc.detail = pb; oc.consumed = 1; return 0; }
        if (R.in_len < 13) { oc.status = SEMZ_EOF; oc.consumed = R.in_len; return 0; }
        uint32_t dict = 0;
        uint64_t size = 0;
        for (int i = 0; i < 4; i++) dict |= (uint32_t)R.in[1 + i] << (8 * i);
        for (int i = 0; i < 8; i++) size |= (uint64_t)R.in[5 + i] << (8 * i);
        const bool have = size != ~0ull;
        if (info) {
            uint32_t *h = info->first_header;                // (:lc :lp :pb :window-size :decompressed-size)
            h[0] = lc; h[1] = lp; h[2] = pb; h[3] = dict; h[4] = have; h[5] = (uint32_t)size; h[6] = (uint32_t)(size >> 32);
        }
        u.in_off = 13;
        u.param0 = dict;
        u.param1 = lc | lp << 8 | pb << 16;
        u.param2 = 0;                                        // :eof-mode :maybe
        if (have && R.c->lz_honour_size) { u.flags |= SEMZ_UF_HAVE_OUT_SIZE; u.size_out = size; }
    } else {
        semz_ctx *c = R.c;
        if (c->lz_lc < 0 || c->lz_lp < 0 || c->lz_pb < 0)
            return semz_fail(c, -1, ":raw-lzma needs the options lc, lp, pb and window-size (lzma.lisp:818-821)");
        u.param0 = (uint32_t)(c->window_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : c->window_size);
        u.param1 = (uint32_t)c->lz_lc | (uint32_t)c->lz_lp << 8 | (uint32_t)c->lz_pb << 16;
        u.param2 = (uint32_t)c->lz_eof_mode;
        if (c->lz_size >= 0) { u.flags |= SEMZ_UF_HAVE_OUT_SIZE; u.size_out = (uint64_t)c->lz_size; }
    }
    const uint32_t lclp = (u.param1 & 0xFF) + ((u.param1 >> 8) & 0xFF);
    u.kind = lclp > 4 ? SEMZ_UNIT_LZMA_WIDE : SEMZ_UNIT_LZMA;
    // raw LZMA has no chunk limit: a match of 273 bytes costs a fraction of a bit, so only memory bounds the
    // output; a declared size caps the span but never sizes the first attempt by itself
    uint64_t limit = ~0ull >> 3;
    if (u.flags & SEMZ_UF_HAVE_OUT_SIZE) {
        limit = std::max<uint64_t>(u.size_out, 1);
        guess = std::min(limit, std::max<uint64_t>(guess, sat_mul(64, R.in_len)));
    }
    semz_result r{};
    int rc = R.decode_one(u, r, guess, limit);
    if (rc) return rc;
    oc.consumed = (r.in_end_bit + 7) / 8;
    if (r.status != SEMZ_OK) { oc.status = r.status; oc.detail = r.detail; return 0; }
    R.emit(r.out_off, r.out_len);
    R.member_done(oc.consumed);
    if (info) info->nmembers = 1;
    if (all_members && oc.consumed != R.in_len) oc.status = SEMZ_E_TRAILING_DATA;       // interface.lisp:125-130
    return 0;
}

}  // namespace

struct semz_decoded { Run run; };

extern "C" void semz_stream_release(semz_decoded *d)
{
    if (!d) return;
    if (semz_ctx *c = d->run.c) {               // (a stream that outlived its context holds host memory only)
        std::lock_guard<std::mutex> g(c->mu);
        for (size_t i = 0; i < c->live.size(); i++)
            if (c->live[i] == d) { c->live.erase(c->live.begin() + i); break; }
    }
    delete d;
}

void semz_stream_orphan_all(semz_ctx *c)
{
    std::vector<semz_decoded *> live;
    {
        std::lock_guard<std::mutex> g(c->mu);
        live.swap(c->live);
    }
    for (semz_decoded *d : live) d->run.free_device();
}

// `sink`: the buffer the caller wants the stream in (semz_decode_stream_into), which the batch loop fills directly
static int stream_decode(semz_ctx *c, int fmt, const uint8_t *in, size_t in_len, int all_members, uint8_t *sink, size_t sink_cap,
                         semz_decoded **decoded, semz_result *res, semz_stream_info *info)
{
    if (!c) return -1;
    if (!decoded || !res || (!in && in_len)) return semz_fail(c, -1, "null pointer");
    *decoded = nullptr;
    memset(res, 0, sizeof *res);
    semz_stream_info si;
    memset(&si, 0, sizeof si);
    // SEMZ_TRACE=1 prints where the call's wall time goes
    const bool trace = getenv("SEMZ_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    semz_decoded *Rp = new semz_decoded{Run{c, in, in_len}};
    Run &R = Rp->run;
    R.sink = sink;
    R.sink_cap = sink_cap;
    // (the upload from pageable memory keeps this thread busy: the scan for gzip member starts runs beside it)
    std::thread scanner;
    if (fmt == SEMZ_FMT_GZIP && all_members && in_len >= (1u << 20))
        scanner = std::thread([&R, in, in_len] { R.cand = scan_gzip_candidates(in, in_len); });
    else if (fmt == SEMZ_FMT_BZIP2 && in_len >= (1u << 20))
        scanner = std::thread([&R, in, in_len] { R.bz_cand = scan_bzip2(in, in_len); });
    int rc = R.open();
    if (scanner.joinable()) {
        scanner.join();
        R.cand_ready = true;
    }
    if (rc) { delete Rp; return rc; }
    const double t1 = now();
    Outcome oc;
    switch (fmt) {
    case SEMZ_FMT_DEFLATE:
    case SEMZ_FMT_ZLIB:
    case SEMZ_FMT_GZIP: {
        // header information of the first member (interface.lisp:118-122)
        semz_unit *su = nullptr;
        size_t sn = 0;
        static const uint64_t zero = 0;
        if (semz_split(fmt, in, in_len, &zero, nullptr, 1, &su, &sn, &si) == 0) free(su);
        rc = run_deflate_family(R, fmt, all_members != 0, oc, &si);
        break;
    }
    case SEMZ_FMT_BZIP2: rc = run_bzip2(R, all_members != 0, oc, &si); break;
    case SEMZ_FMT_XZ: rc = run_xz(R, all_members != 0, oc, &si); break;
    case SEMZ_FMT_RAW_LZMA:
    case SEMZ_FMT_LZMA: rc = run_lzma(R, fmt, all_members != 0, oc, &si); break;
    case SEMZ_FMT_RAW_LZMA2:
    case SEMZ_FMT_LZMA2: rc = run_lzma2(R, fmt, all_members != 0, oc, &si); break;
    default: rc = semz_fail(c, -1, "format %d has no stream decoder", fmt); break;
    }
    if (rc) { delete Rp; return rc; }
    const double t2 = now();
    c->last_windows = R.windows_run;
    c->last_uploaded = R.uploaded.load();
    R.drop_working_set();
    if (trace) fprintf(stderr, "semz_stream_decode: open+upload %.1f ms, split+decode %.1f ms, drop %.1f ms, %zu windows\n", t1 - t0, t2 - t1, now() - t2, R.windows_run);
    res->status = oc.status;
    res->detail = oc.detail;
    res->in_end_bit = 8 * oc.consumed;
    res->out_len = R.emitted;
    if (oc.status != SEMZ_OK) {                 // the other arguments of the reference's message (see Outcome)
        res->aux = oc.aux;
        memcpy(res->check, oc.check, sizeof res->check);
        res->check_len = oc.check_len;
    }
    si.total_out = R.emitted;
    if (info) *info = si;
    {
        std::lock_guard<std::mutex> g(c->mu);
        c->live.push_back(Rp);
    }
    *decoded = Rp;
    return 0;
}

extern "C" int semz_stream_decode(semz_ctx *c, int fmt, const uint8_t *in, size_t in_len, int all_members,
                                  semz_decoded **decoded, semz_result *res, semz_stream_info *info)
{
    return stream_decode(c, fmt, in, in_len, all_members, nullptr, 0, decoded, res, info);
}

extern "C" int semz_stream_members(const semz_decoded *d, const semz_member **members, size_t *nmembers)
{
    if (!d || !members || !nmembers) return -1;
    *members = d->run.members.data();
    *nmembers = d->run.members.size();
    return 0;
}

extern "C" int semz_stream_read(semz_decoded *d, uint64_t offset, void *dst, size_t len)
{
    if (!d) return -1;
    semz_ctx *c = d->run.c;
    if (!c) return -6;                          // the context was closed: the decoded data went with it
    if (!dst && len) return semz_fail(c, -1, "null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    return d->run.read(offset, (uint8_t *)dst, len);
}

extern "C" int semz_decode_stream_into(semz_ctx *c, int fmt, const uint8_t *in, size_t in_len, int all_members,
                                       uint8_t *out, size_t out_cap, size_t *out_len, semz_result *res,
                                       semz_stream_info *info)
{
    if (!c) return -1;
    if (!out_len || (!out && out_cap)) return semz_fail(c, -1, "null pointer");
    *out_len = 0;
    semz_decoded *d = nullptr;
    int rc = stream_decode(c, fmt, in, in_len, all_members, out, out_cap, &d, res, info);
    if (rc) return rc;
    Run *R = &d->run;
    *out_len = R->emitted;
    if (R->emitted > out_cap) {                          // the caller learns the size and decodes again, or uses the stepwise calls
        semz_stream_release(d);
        return 1;
    }
    const bool trace = getenv("SEMZ_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    rc = R->read(0, out, R->emitted);
    if (trace) fprintf(stderr, "semz_decode_stream_into: download %.1f ms\n", now() - t0);
    semz_stream_release(d);
    return rc;
}

extern "C" int semz_decode_stream(semz_ctx *c, int fmt, const uint8_t *in, size_t in_len, int all_members, uint8_t **out,
                                  size_t *out_len, semz_result *res, semz_stream_info *info)
{
    if (!c) return -1;
    if (!out || !out_len || !res || (!in && in_len)) return semz_fail(c, -1, "null pointer");
    *out = nullptr;
    *out_len = 0;
    semz_decoded *d = nullptr;
    int rc = semz_stream_decode(c, fmt, in, in_len, all_members, &d, res, info);
    if (rc) return rc;
    const bool trace = getenv("SEMZ_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    rc = d->run.gather(out, out_len);
    if (trace) fprintf(stderr, "semz_decode_stream: download %.1f ms\n", now() - t0);
    semz_stream_release(d);
    return rc;
}
