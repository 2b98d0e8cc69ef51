// CPU-side check of the multi-device partition / merge logic (decompress_amd/csrc/multi.h) with made-up
// per-device results: no HIP, built with g++ by tests/test_multi_partition.py.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../decompress_amd/csrc/multi.h"

using namespace semz;

static semz_unit unit(uint64_t in_off, uint64_t in_limit, uint64_t out_off, uint64_t out_cap)
{
    semz_unit u;
    memset(&u, 0, sizeof u);
    u.kind = SEMZ_UNIT_GZIP;
    u.in_off = in_off; u.in_limit = in_limit; u.out_off = out_off; u.out_cap = out_cap;
    return u;
}

// every device "decodes" its localised units: out_len = out_cap, out_off = where it was told to write (or a
// running offset for SEMZ_OUT_AUTO), in_end_bit = 8 * (local in_off + 10); a unit cut short fails when told to
static std::vector<semz_result> fake_device(const std::vector<semz_unit> &local, const std::vector<uint8_t> &clamped, bool fail_cut)
{
    std::vector<semz_result> r(local.size());
    uint64_t auto_off = 0;
    for (size_t i = 0; i < local.size(); i++) {
        memset(&r[i], 0, sizeof r[i]);
        r[i].status = fail_cut && clamped[i] ? SEMZ_EOF : SEMZ_OK;
        r[i].out_len = local[i].out_cap ? local[i].out_cap : 1000;
        if (local[i].out_off == SEMZ_OUT_AUTO) { r[i].out_off = auto_off; auto_off += r[i].out_len; }
        else r[i].out_off = local[i].out_off;
        r[i].in_end_bit = 8 * (local[i].in_off + 10);
    }
    return r;
}

static void check(size_t n, unsigned parts, bool fixed, bool fail_cut)
{
    const uint64_t in_len = 1000 * n + 77;
    std::vector<semz_unit> u;
    uint64_t out = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t cap = 100 + 7 * (i % 13);
        u.push_back(unit(1000 * i + (i % 3), in_len, fixed ? out : SEMZ_OUT_AUTO, fixed ? cap : 0));
        out += cap;
    }
    const uint64_t frame_base = 4096;
    auto ranges = partition_units(u.data(), n, in_len, parts, 64, 1000 * n);
    assert(!ranges.empty() && ranges.size() <= parts);
    // contiguous, complete, in order; byte-balanced within one unit
    size_t at = 0;
    for (auto &r : ranges) { assert(r.first == at && r.count > 0); at += r.count; }
    assert(at == n);
    if (ranges.size() == parts && n >= 4 * parts)
        for (auto &r : ranges) assert(r.count + 2 >= n / parts && r.count <= n / parts + 2);
    // no device -- the primary (range 0) included -- is handed more of the input than its own units' bytes and the
    // margin: together the devices upload the input once, plus a margin per cut
    uint64_t uploaded = 0;
    for (size_t p = 0; p < ranges.size(); p++) {
        const MultiRange &r = ranges[p];
        const uint64_t own_hi = p + 1 < ranges.size() ? u[ranges[p + 1].first].in_off : in_len;
        assert(r.in_hi - r.in_lo <= (own_hi - r.in_lo) + 64);
        if (ranges.size() > 1) assert(r.in_hi - r.in_lo < in_len);
        uploaded += r.in_hi - r.in_lo;
    }
    assert(uploaded <= in_len + 64 * ranges.size());
    std::vector<semz_result> res(n);
    std::vector<uint8_t> distrust(n, 0xAA);
    for (size_t p = 0; p < ranges.size(); p++) {
        const MultiRange &r = ranges[p];
        assert(r.in_lo == u[r.first].in_off);
        assert(r.in_hi <= in_len && r.in_hi > u[r.first + r.count - 1].in_off);
        if (p + 1 < ranges.size()) assert(r.in_hi >= u[ranges[p + 1].first].in_off);       // the whole range is inside its bytes
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(u.data(), r, local, clamped);
        for (size_t i = 0; i < r.count; i++) {
            assert(local[i].in_off == u[r.first + i].in_off - r.in_lo);
            assert(local[i].in_limit <= r.in_hi - r.in_lo && local[i].in_off <= local[i].in_limit);
            assert(clamped[i] == (u[r.first + i].in_limit > r.in_hi));
            if (fixed) assert(local[i].out_off + local[i].out_cap <= r.out_room);
        }
        auto lres = fake_device(local, clamped, fail_cut);
        const uint64_t dev_base = 512 * p;                   // the device's arena had 512 p bytes in it already
        globalise_results(r, (unsigned)p, dev_base, frame_base, clamped, lres.data(), res.data(), distrust.data());
    }
    // merged: input positions are global again, output offsets name device and place once the caller adds its base
    for (size_t p = 0; p < ranges.size(); p++)
        for (size_t i = ranges[p].first; i < ranges[p].first + ranges[p].count; i++) {
            assert(res[i].in_end_bit == 8 * (u[i].in_off + 10));
            const uint64_t off = res[i].out_off + (p ? frame_base : 0);
            assert(multi_dev(off) == p);
            if (fixed && p) assert(multi_off(off) == 512 * p + (u[i].out_off - ranges[p].out_lo));
            if (fixed && !p) assert(off == u[i].out_off);
            const bool cut = u[i].in_limit > ranges[p].in_hi;
            assert(distrust[i] == (uint8_t)(fail_cut && cut));
        }
}

// Ranges are balanced by output bytes where every unit's size is known (fixed spans), by compressed bytes otherwise:
// units whose output sizes differ a hundredfold while their compressed sizes are alike.
static void check_balance()
{
    const size_t n = 400;
    const uint64_t in_len = 1000 * n;
    for (int fixed = 0; fixed < 2; fixed++) {
        std::vector<semz_unit> u;
        uint64_t out = 0, total = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t cap = i < n / 4 ? 100000 : 1000;           // the first quarter holds most of the output
            u.push_back(unit(1000 * i, in_len, fixed ? out : SEMZ_OUT_AUTO, fixed ? cap : 0));
            out += cap;
            total += cap;
        }
        auto ranges = partition_units(u.data(), n, in_len, 4, 64, total);
        assert(ranges.size() == 4);
        if (fixed) {
            for (auto &r : ranges) {
                uint64_t w = 0;
                for (size_t i = r.first; i < r.first + r.count; i++) w += u[i].out_cap;
                assert(w + 100000 >= total / 4 && w <= total / 4 + 100000);
            }
            assert(ranges[0].count < n / 8);                          // far fewer units where the units are large
        } else {
            for (auto &r : ranges) assert(r.count == n / 4);
        }
    }
    // sizes that are claims of the data and do not add up: back to compressed bytes, no overflow
    std::vector<semz_unit> u;
    for (size_t i = 0; i < 8; i++) u.push_back(unit(1000 * i, 8000, 0, ~0ull / 2));
    auto ranges = partition_units(u.data(), 8, 8000, 4, 64, 0);
    assert(ranges.size() == 4);
    for (auto &r : ranges) assert(r.count == 2);
}

// the batch loop's windows: contiguous, complete, every one within the limit unless it is a single unit, and the
// merge with window tags (device 255, window number in the offset's high bits) names window and place
static void check_windows(size_t n, uint64_t limit, bool fixed, uint64_t unit_scratch)
{
    const uint64_t in_len = 1000 * n + 77, margin = 64;
    std::vector<semz_unit> u;
    std::vector<uint64_t> need;
    uint64_t out = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t cap = 100 + 7 * (i % 13) + (i % 17 == 5 ? 5000 : 0);
        u.push_back(unit(1000 * i + (i % 3), in_len, fixed ? out : SEMZ_OUT_AUTO, fixed ? cap : 0));
        need.push_back(cap);
        out += cap;
    }
    auto wins = window_units(u.data(), n, in_len, need.data(), limit, margin, unit_scratch);
    size_t at = 0;
    std::vector<semz_result> res(n);
    std::vector<uint8_t> distrust(n, 0xAA);
    const uint64_t frame_base = 12345;
    for (size_t w = 0; w < wins.size(); w++) {
        const MultiRange &r = wins[w];
        assert(r.first == at && r.count > 0);
        at += r.count;
        assert(r.in_lo == u[r.first].in_off && r.in_hi <= in_len);
        if (w + 1 < wins.size()) assert(r.in_hi >= u[wins[w + 1].first].in_off);
        uint64_t room = 0;
        for (size_t i = 0; i < r.count; i++) room += need[r.first + i];
        assert(r.out_room == room);                                   // (fixed spans lie back to back here)
        const uint64_t cost = (r.in_hi - r.in_lo) + room + r.count * unit_scratch;
        assert(cost <= limit || r.count == 1);
        // greedy: the next unit would not have fitted
        if (w + 1 < wins.size() && r.count >= 1) {
            const size_t e = r.first + r.count;
            uint64_t h = e + 1 < n ? u[e + 1].in_off : in_len;
            h = h + margin < in_len ? h + margin : in_len;
            assert((h - r.in_lo) + room + need[e] + (r.count + 1) * unit_scratch > limit);
        }
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(u.data(), r, local, clamped);
        auto lres = fake_device(local, clamped, true);
        globalise_results(r, 255, (uint64_t)w << 40, frame_base, clamped, lres.data(), res.data(), distrust.data());
    }
    assert(at == n);
    for (size_t w = 0; w < wins.size(); w++)
        for (size_t i = wins[w].first; i < wins[w].first + wins[w].count; i++) {
            const uint64_t off = res[i].out_off + frame_base;
            assert(multi_dev(off) == 255 && (multi_off(off) >> 40) == w);
            if (fixed) assert((multi_off(off) & ((1ull << 40) - 1)) == u[i].out_off - wins[w].out_lo);
            assert(res[i].in_end_bit == 8 * (u[i].in_off + 10));
            assert(distrust[i] == (uint8_t)(u[i].in_limit > wins[w].in_hi));
        }
}

int main()
{
    check_balance();
    for (size_t n : {1, 2, 3, 7, 64, 1000})
        for (uint64_t limit : {1ull, 3000ull, 20000ull, 1ull << 40})
            for (int fixed = 0; fixed < 2; fixed++)
                for (uint64_t scratch : {0ull, 700ull}) check_windows(n, limit, fixed != 0, scratch);
    for (size_t n : {1, 2, 3, 7, 64, 1000})
        for (unsigned parts : {1u, 2u, 3u, 8u})
            for (int fixed = 0; fixed < 2; fixed++)
                for (int fail = 0; fail < 2; fail++) check(n, parts, fixed != 0, fail != 0);
    // units with a known end (xz blocks): a range's bytes reach to the end of its last unit
    {
        std::vector<semz_unit> u;
        for (int i = 0; i < 10; i++) u.push_back(unit(100 * i, 100 * i + 130, 50 * i, 50));
        auto r = partition_units(u.data(), u.size(), 5000, 2, 0, 0);
        assert(r.size() == 2 && r[0].in_hi >= u[r[0].count - 1].in_limit);
        std::vector<semz_unit> local;
        std::vector<uint8_t> clamped;
        localise_units(u.data(), r[0], local, clamped);
        for (auto c : clamped) assert(!c);
    }
    puts("multi partition / merge: ok");
    return 0;
}
