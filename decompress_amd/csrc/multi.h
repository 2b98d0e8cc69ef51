// multi.h -- one input, several GPUs: the partition of a batch of units and the merge of the per-device results.
//
// The path shards by unit and has no exchange step: the units of one host-side split are
// cut into contiguous ranges of about equal work (output bytes where they are known, compressed bytes otherwise),
// every range is decoded on a device of its own -- the primary device included -- from the bytes of that range only, and the results come back into the frame of the whole batch, where the host's
// cross-unit walk (stream CRC fold bzip2.lisp:42-49, xz index xz.lisp:186-203, first error wins
// interface.lisp:91-133) reads them as if one device had produced them.  These functions are the pure part of
// that (no HIP): stream.cpp calls them, tests/cpp/test_multi.cpp checks them with made-up per-device results.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include "../../include/semz_gpu.h"

namespace semz {

// Where a device's results live: arena offsets of helper devices carry the device's index in their top byte
// (the primary device is 0, so its offsets are plain).
constexpr int MULTI_TAG_SHIFT = 56;
constexpr uint64_t MULTI_OFF_MASK = (1ull << MULTI_TAG_SHIFT) - 1;
inline uint64_t multi_tag(unsigned dev, uint64_t off) { return ((uint64_t)dev << MULTI_TAG_SHIFT) | off; }
inline unsigned multi_dev(uint64_t off) { return (unsigned)(off >> MULTI_TAG_SHIFT); }
inline uint64_t multi_off(uint64_t off) { return off & MULTI_OFF_MASK; }

struct MultiRange {
    size_t first, count;          // units [first, first + count)
    uint64_t in_lo, in_hi;        // bytes of the input this range is decoded from: [in_lo, in_hi)
    bool fixed;                   // the units carry their own output offsets (not SEMZ_OUT_AUTO)
    uint64_t out_lo, out_room;    // fixed: the span of the batch's output the range covers; auto: room asked for
};

// Contiguous ranges of units (units are in input order), balanced by what a device's time goes with: the output
// bytes where every unit's size is known before it is decoded (fixed spans: gzip ISIZE gzip.lisp:109, the xz index
// xz.lisp:186-187, caller-supplied record sizes), the compressed bytes between unit starts otherwise (bzip2
// blocks: SEMZ_OUT_AUTO) -- SURVEY.md section 8e.  `margin` bytes behind the last unit's span are part of a
// range's input (decoders stage their input in pieces and bzip2 blocks end inside a byte).  Fewer ranges than
// `parts` come back when there are not enough units.
inline std::vector<MultiRange> partition_units(const semz_unit *u, size_t n, size_t in_len, unsigned parts, uint64_t margin,
                                               uint64_t auto_room)
{
    std::vector<MultiRange> out;
    if (!n || !parts) return out;
    if (parts > n) parts = (unsigned)n;
    const uint64_t lo = u[0].in_off;
    const uint64_t span = in_len > lo ? in_len - lo : 0;
    // weight of unit i and the running sum in front of it
    bool by_output = true;
    for (size_t i = 0; i < n && by_output; i++) by_output = u[i].out_off != SEMZ_OUT_AUTO;
    uint64_t total = 0;
    for (size_t i = 0; i < n && by_output; i++) {
        if (__builtin_add_overflow(total, u[i].out_cap, &total)) by_output = false;      // (sizes are claims of the data)
    }
    if (by_output && !total) by_output = false;
    if (!by_output) total = span;
    uint64_t before = 0;                                            // weight of the units in front of `end`
    size_t at = 0;
    for (unsigned p = 0; p < parts && at < n; p++) {
        // the range ends where the running weight passes its share
        const uint64_t goal = (uint64_t)((unsigned __int128)total * (p + 1) / parts);
        auto weight_before = [&](size_t i) { return by_output ? before : u[i].in_off - lo; };
        size_t end = at + 1;
        if (by_output) before += u[at].out_cap;
        if (p + 1 == parts) end = n;
        else
            while (end < n && n - end > parts - 1 - p && weight_before(end) < goal) {
                if (by_output) before += u[end].out_cap;
                end++;
            }
        MultiRange r{};
        r.first = at;
        r.count = end - at;
        r.in_lo = u[at].in_off;
        uint64_t hi = end < n ? u[end].in_off : in_len;
        for (size_t i = at; i < end; i++)                       // a unit with a known end may reach behind the next start
            if (u[i].in_limit < in_len && u[i].in_limit > hi) hi = u[i].in_limit;
        hi = hi + margin < in_len ? hi + margin : in_len;
        r.in_hi = hi;
        r.fixed = u[at].out_off != SEMZ_OUT_AUTO;
        if (r.fixed) {
            uint64_t a = ~0ull, b = 0;
            for (size_t i = at; i < end; i++) {
                if (u[i].out_off < a) a = u[i].out_off;
                if (u[i].out_off + u[i].out_cap > b) b = u[i].out_off + u[i].out_cap;
            }
            r.out_lo = a;
            r.out_room = b - a;
        } else {
            r.out_lo = 0;
            r.out_room = auto_room / n * r.count + auto_room / n + (1u << 20);   // its share and a little
        }
        out.push_back(r);
        at = end;
    }
    return out;
}

// The batch loop: contiguous ranges again, but as many as it takes for every range to fit
// `limit` bytes of device memory -- the range's input bytes, its output room and `unit_scratch` bytes per unit of
// decoder scratch.  `out_need[i]` is the room unit i asks for (its span if fixed, an estimate if the device packs
// the outputs).  A range holds at least one unit, whatever that unit costs.
inline std::vector<MultiRange> window_units(const semz_unit *u, size_t n, size_t in_len, const uint64_t *out_need, uint64_t limit,
                                            uint64_t margin, uint64_t unit_scratch)
{
    std::vector<MultiRange> out;
    size_t at = 0;
    while (at < n) {
        MultiRange r{};
        r.first = at;
        r.in_lo = u[at].in_off;
        r.fixed = u[at].out_off != SEMZ_OUT_AUTO;
        uint64_t room = 0, a = ~0ull, b = 0, hi = 0;
        size_t end = at;
        while (end < n) {
            uint64_t h = end + 1 < n ? u[end + 1].in_off : in_len;
            if (u[end].in_limit < in_len && u[end].in_limit > h) h = u[end].in_limit;
            if (h < hi) h = hi;
            const uint64_t h_m = h + margin < in_len ? h + margin : in_len;
            const uint64_t cost = (h_m - r.in_lo) + room + out_need[end] + (uint64_t)(end + 1 - at) * unit_scratch;
            if (end > at && cost > limit) break;
            hi = h;
            room += out_need[end];
            if (r.fixed) {
                if (u[end].out_off < a) a = u[end].out_off;
                if (u[end].out_off + u[end].out_cap > b) b = u[end].out_off + u[end].out_cap;
            }
            end++;
        }
        r.count = end - at;
        r.in_hi = hi + margin < in_len ? hi + margin : in_len;
        r.out_lo = r.fixed ? a : 0;
        r.out_room = r.fixed ? b - a : room;
        out.push_back(r);
        at = end;
    }
    return out;
}

// A range's units in the frame of the device that decodes them: input offsets from in_lo, limits cut at in_hi,
// fixed output offsets from out_lo.  `clamped[i]` says that unit i lost part of its input span.
inline void localise_units(const semz_unit *u, const MultiRange &r, std::vector<semz_unit> &local, std::vector<uint8_t> &clamped)
{
    local.assign(u + r.first, u + r.first + r.count);
    clamped.assign(r.count, 0);
    for (size_t i = 0; i < r.count; i++) {
        semz_unit &x = local[i];
        if (x.in_limit > r.in_hi) { x.in_limit = r.in_hi; clamped[i] = 1; }
        x.in_off -= r.in_lo;
        x.in_limit -= r.in_lo;
        if (x.out_off != SEMZ_OUT_AUTO) x.out_off -= r.out_lo;
    }
}

// A device's results back in the frame of the batch.  `frame_base` is what the caller of the batch adds to every
// out_off afterwards (its arena base on the primary device): results of a helper device (dev > 0) are stored so
// that this addition yields the tagged offset.  A failed unit that was decoded from a cut span proves nothing
// (the failure may be the cut): distrust[i] = 1 tells the walk to decode it again where the whole input is.
inline void globalise_results(const MultiRange &r, unsigned dev, uint64_t dev_arena_base, uint64_t frame_base,
                              const std::vector<uint8_t> &clamped, const semz_result *local, semz_result *res, uint8_t *distrust)
{
    for (size_t i = 0; i < r.count; i++) {
        semz_result x = local[i];
        x.in_end_bit += 8 * r.in_lo;
        if (dev == 0) x.out_off += r.fixed ? r.out_lo : 0;
        else x.out_off = multi_tag(dev, dev_arena_base + x.out_off) - frame_base;
        res[r.first + i] = x;
        distrust[r.first + i] = (uint8_t)(clamped[i] && x.status != SEMZ_OK);
    }
}

}  // namespace semz
