// split.cpp -- host splitter: finds the independent units of a multi-unit input
// without decoding it (include/semz_gpu.h semz_split).
#include <array>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "container.h"

namespace semz {

uint32_t crc32_host(const uint8_t *p, size_t n)          // crc.lisp:19-110
{
    static const std::array<uint32_t, 256> table = [] {     // initialised once, thread-safe (C++11 static)
        std::array<uint32_t, 256> t{};
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
            t[i] = c;
        }
        return t;
    }();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}

// ---- gzip header (gzip.lisp:31-87) ----
GzipHeader parse_gzip_header(const uint8_t *in, uint64_t pos, uint64_t limit)
{
    GzipHeader h;
    uint64_t p = pos;
    auto fail = [&](int st, uint32_t d) { h.status = st; h.detail = d; h.header_len = p - pos; return h; };
    if (limit - p < 10) return fail(SEMZ_EOF, 0);
    const uint32_t flg = in[p + 3];
    if (in[p] != 0x1F || in[p + 1] != 0x8B) return fail(SEMZ_E_GZIP_MAGIC, 0);
    if (in[p + 2] != 8) return fail(SEMZ_E_GZIP_METHOD, in[p + 2]);
    if (flg >> 5) return fail(SEMZ_E_GZIP_RESERVED_FLAGS, 0);
    h.textp = flg & 1;
    h.mtime = (uint32_t)in[p + 4] | ((uint32_t)in[p + 5] << 8) | ((uint32_t)in[p + 6] << 16) | ((uint32_t)in[p + 7] << 24);
    h.xfl = in[p + 8];
    h.os = in[p + 9];
    p += 10;
    if (flg & 4) {
        if (limit - p < 2) return fail(SEMZ_EOF, 0);
        const uint64_t xl = in[p] | ((uint32_t)in[p + 1] << 8);
        p += 2;
        if (limit - p < xl) return fail(SEMZ_EOF, 0);
        uint64_t i = 0;                                   // gzip-extra-fields-consistent-p, gzip.lisp:10-25
        while (i < xl) {
            if (i + 4 > xl) return fail(SEMZ_E_GZIP_EXTRA_FIELDS, 0);
            const uint64_t l = in[p + i + 2] | ((uint32_t)in[p + i + 3] << 8);
            if (i + 4 + l > xl) return fail(SEMZ_E_GZIP_EXTRA_FIELDS, 0);
            if (in[p + i] == 'B' && in[p + i + 1] == 'C' && l == 2)
                h.bgzf_size = (int64_t)(in[p + i + 4] | ((uint32_t)in[p + i + 5] << 8)) + 1;
            i += 4 + l;
        }
        h.extra_off = (int64_t)p;
        h.extra_len = (int64_t)xl;
        p += xl;
    }
    for (int which = 0; which < 2; which++) {
        if (!(flg & (which ? 16 : 8))) continue;
        const uint64_t s = p;
        for (;;) {
            if (p >= limit) return fail(SEMZ_EOF, 0);
            if (in[p++] == 0) break;
        }
        if (which) { h.comment_off = (int64_t)s; h.comment_len = (int64_t)(p - 1 - s); }
        else { h.name_off = (int64_t)s; h.name_len = (int64_t)(p - 1 - s); }
    }
    if (flg & 2) {
        if (limit - p < 2) return fail(SEMZ_EOF, 0);
        const uint32_t want = in[p] | ((uint32_t)in[p + 1] << 8);
        const uint32_t got = crc32_host(in + pos, p - pos) & 0xFFFF;
        p += 2;
        if (want != got) return fail(SEMZ_E_GZIP_HEADER_CRC, want);
    }
    h.header_len = p - pos;
    return h;
}

// ---- bzip2 ----
uint64_t read_bits_be(const uint8_t *in, uint64_t in_len, uint64_t bit, unsigned n)
{
    uint64_t v = 0;
    for (unsigned i = 0; i < n; i++) {
        const uint64_t b = bit + i;
        const unsigned x = (b >> 3) < in_len ? (in[b >> 3] >> (7 - (b & 7))) & 1 : 0;
        v = (v << 1) | x;
    }
    return v;
}

int parse_bzip2_header(const uint8_t *in, uint64_t pos, uint64_t in_len, uint32_t *level, uint32_t *detail)
{
    *detail = 0;
    if (in_len - pos < 2) return SEMZ_EOF;
    if (in[pos] != 0x42 || in[pos + 1] != 0x5A) { *detail = ((uint32_t)in[pos] << 8) | in[pos + 1]; return SEMZ_E_BZ_MAGIC; }
    if (in_len - pos < 4) return SEMZ_EOF;
    if (in[pos + 2] != 0x68) { *detail = in[pos + 2]; return SEMZ_E_BZ_VERSION; }
    if (in[pos + 3] < 0x31 || in[pos + 3] > 0x39) { *detail = in[pos + 3]; return SEMZ_E_BZ_LEVEL; }
    *level = in[pos + 3] - 0x30;
    return SEMZ_OK;
}

std::vector<BzCandidate> scan_bzip2(const uint8_t *in, uint64_t in_len)
{
    // A 64 Ki-entry filter on the last two bytes of the window rejects 99 % of the positions before the eight
    // shifted 48-bit compares; the input is scanned in slices by a few host threads.
    static const std::vector<uint8_t> maybe = [] {          // initialised once, thread-safe (C++11 static)
        std::vector<uint8_t> m(65536, 0);
        for (unsigned s = 0; s < 8; s++)
            for (uint64_t magic : {BZ_BLOCK_MAGIC, BZ_FOOTER_MAGIC})
                for (unsigned low = 0; low < (1u << s); low++)
                    m[(uint16_t)((magic << s) | low)] = 1;
        return m;
    }();
    const unsigned nthreads = in_len >= (8u << 20) ? 8 : 1;
    std::vector<std::vector<BzCandidate>> parts(nthreads);
    auto scan = [&](unsigned t) {
        const uint64_t lo = in_len * t / nthreads, hi = in_len * (t + 1) / nthreads;
        std::vector<BzCandidate> &out = parts[t];
        for (uint64_t i = lo < 5 ? 5 : lo; i < hi; i++) {
            if (!maybe[(uint16_t)((uint32_t)in[i - 1] << 8 | in[i])]) continue;
            uint64_t w = 0;                                   // bytes i-7..i, big-endian (zero before the start)
            for (int k = 7; k >= 0; k--) w = (w << 8) | (i >= (uint64_t)k ? in[i - k] : 0);
            // a 48-bit pattern that ends s bits (0..7) before the end of byte i; higher s first = lower bit position
            for (int s = 7; s >= 0; s--) {
                const uint64_t v = (w >> s) & 0xFFFFFFFFFFFFull;
                if (v != BZ_BLOCK_MAGIC && v != BZ_FOOTER_MAGIC) continue;
                if (i < 6 && s != 0) continue;                // needs 7 bytes unless byte aligned
                out.push_back(BzCandidate{8 * (i + 1) - (unsigned)s - 48, v == BZ_FOOTER_MAGIC, 9});
            }
        }
    };
    if (nthreads == 1) scan(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; t++) th.emplace_back(scan, t);
        for (auto &x : th) x.join();
    }
    std::vector<BzCandidate> out;
    uint32_t level = 9, det;
    if (parse_bzip2_header(in, 0, in_len, &level, &det) != SEMZ_OK) level = 9;
    for (auto &p : parts)
        for (auto c : p) {
            c.level = level;
            out.push_back(c);
            if (c.footer) {
                // a following stream header, if any, sets the level of the next blocks
                const uint64_t next = (c.bit + 48 + 32 + 7) >> 3;
                uint32_t lv;
                if (next + 4 <= in_len && parse_bzip2_header(in, next, in_len, &lv, &det) == SEMZ_OK) level = lv;
            }
        }
    return out;
}

// ---- xz ----
unsigned xz_check_size(uint32_t t)
{
    static const uint8_t sizes[16] = {0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64};
    return sizes[t & 15];
}

int xz_varint(const uint8_t *in, uint64_t *pos, uint64_t limit, uint64_t *value)
{
    if (*pos >= limit) return SEMZ_EOF;
    const uint32_t x0 = in[(*pos)++];
    if (x0 < 0x80) { *value = x0; return SEMZ_OK; }
    uint64_t v = x0 & 0x7F;
    for (int i = 7; i < 63; i += 7) {
        if (*pos >= limit) return SEMZ_EOF;
        const uint32_t b = in[(*pos)++];
        v |= (uint64_t)(b & 0x7F) << i;
        if (b == 0) return SEMZ_E_XZ_VARINT_ZERO;
        if (!(b & 0x80)) { *value = v; return SEMZ_OK; }
    }
    return SEMZ_E_XZ_VARINT_LONG;
}

int parse_xz_stream_header(const uint8_t *in, uint64_t pos, uint64_t in_len, bool skip_magic, uint32_t *check_type,
                           uint64_t *end, uint32_t *detail)
{
    static const uint8_t magic[6] = {0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00};
    *detail = 0;
    *end = pos;
    if (!skip_magic) {
        if (in_len - pos < 6) { *end = in_len; return SEMZ_EOF; }
        if (memcmp(in + pos, magic, 6)) { *end = pos + 6; return SEMZ_E_XZ_MAGIC; }
        pos += 6;
    }
    *end = pos;
    if (pos >= in_len) return SEMZ_EOF;
    if (in[pos] != 0) { *end = pos + 1; return SEMZ_E_XZ_RESERVED_BYTE; }
    if (in_len - pos < 2) { *end = in_len; return SEMZ_EOF; }
    const uint32_t h2 = in[pos + 1];
    if (in_len - pos < 6) { *end = in_len; return SEMZ_EOF; }
    *end = pos + 6;
    const uint32_t crc = (uint32_t)in[pos + 2] | ((uint32_t)in[pos + 3] << 8) | ((uint32_t)in[pos + 4] << 16) | ((uint32_t)in[pos + 5] << 24);
    const uint8_t hb[2] = {0, (uint8_t)h2};
    if (crc != crc32_host(hb, 2)) return SEMZ_E_XZ_HEADER_CRC;
    if (h2 >> 4) return SEMZ_E_XZ_RESERVED_FLAGS;
    const uint32_t t = h2 & 15;
    if (!(t == 0 || t == 1 || t == 4 || t == 10)) { *detail = t; return SEMZ_E_XZ_CHECK_TYPE; }
    *check_type = t;
    return SEMZ_OK;
}

XzBlockHeader parse_xz_block_header(const uint8_t *in, uint64_t pos, uint64_t in_len)
{
    XzBlockHeader h;
    auto fail = [&](int st, uint32_t d) { h.status = st; h.detail = d; return h; };
    const uint64_t hlen = 4ull * in[pos];
    h.header_size = hlen;
    if (in_len - pos < hlen + 4) return fail(SEMZ_EOF, 0);
    const uint8_t *hdr = in + pos;
    const uint32_t crc = (uint32_t)hdr[hlen] | ((uint32_t)hdr[hlen + 1] << 8) | ((uint32_t)hdr[hlen + 2] << 16) | ((uint32_t)hdr[hlen + 3] << 24);
    if (crc != crc32_host(hdr, hlen)) { h.detail2 = crc32_host(hdr, hlen); return fail(SEMZ_E_XZ_BLOCK_HDR_CRC, crc); }
    // handle-xz-block, xz.lisp:260-379
    const uint32_t flags = hdr[1], nfilters = 1 + (flags & 3);
    if ((flags >> 2) & 15) return fail(SEMZ_E_XZ_BLOCK_RESERVED, 0);
    h.have_csize = (flags >> 6) & 1;
    h.have_usize = (flags >> 7) & 1;
    uint64_t p = 2, fid[4], fplen[4];
    const uint8_t *fprops[4];
    auto vi = [&](uint64_t *v) {
        int st = xz_varint(hdr, &p, hlen, v);
        return st == SEMZ_EOF ? (int)SEMZ_E_XZ_BLOCK_TRUNCATED : st;     // xz.lisp:287-288
    };
    int st;
    if (h.have_csize && (st = vi(&h.csize)) != SEMZ_OK) return fail(st, 0);
    if (h.have_usize && (st = vi(&h.usize)) != SEMZ_OK) return fail(st, 0);
    for (uint32_t f = 0; f < nfilters; f++) {
        if ((st = vi(&fid[f])) != SEMZ_OK) return fail(st, 0);
        if ((st = vi(&fplen[f])) != SEMZ_OK) return fail(st, 0);
        if (fplen[f] > hlen - p) return fail(SEMZ_E_XZ_BLOCK_TRUNCATED, 0);
        fprops[f] = hdr + p;
        p += fplen[f];
    }
    for (uint64_t i = p; i < hlen; i++)
        if (hdr[i]) return fail(SEMZ_E_XZ_BLOCK_HDR_PADDING, 0);
    // filter chain, last filter first (xz.lisp:311-377); decoding applies them in that same order
    uint32_t stage = 0, ndelta = 0;
    bool have_bcj = false;
    for (uint32_t idx = nfilters; idx >= 1; idx--) {
        const uint64_t id = fid[idx - 1], pl = fplen[idx - 1];
        const uint8_t *pp = fprops[idx - 1];
        if (id == 0x03) {
            if (idx == nfilters) return fail(SEMZ_E_XZ_DELTA_LAST, 0);
            if (pl != 1) return fail(SEMZ_E_XZ_DELTA_PROPS, (uint32_t)pl);
            h.filters |= (uint32_t)SEMZ_FILTER_DELTA << (4 * stage++);
            h.delta_dists |= (uint32_t)pp[0] << (8 * ndelta++);
        } else if (id >= 0x04 && id <= 0x0A) {
            static const uint8_t align[7] = {1, 4, 16, 4, 2, 4, 4};
            if (idx == nfilters) return fail(SEMZ_E_XZ_BCJ_LAST, 0);
            uint32_t off = 0;
            if (pl == 4) off = (uint32_t)pp[0] | ((uint32_t)pp[1] << 8) | ((uint32_t)pp[2] << 16) | ((uint32_t)pp[3] << 24);
            else if (pl != 0) return fail(SEMZ_E_XZ_BCJ_PROPS, (uint32_t)pl);
            if (off % align[id - 4]) { h.detail2 = align[id - 4]; return fail(SEMZ_E_XZ_BCJ_ALIGN, off); }
            // semz_unit has one start offset: a second BCJ stage with another offset is not representable
            if (have_bcj && off != h.bcj_offset) return fail(SEMZ_E_UNSUPPORTED, (uint32_t)id);
            have_bcj = true;
            h.bcj_offset = off;
            // xz ids 0x04 x86, 0x05 PowerPC, 0x06 IA-64, 0x07 ARM, 0x08 ARM-Thumb, 0x09 SPARC, 0x0A ARM64 (xz.lisp:333-339)
            static const uint8_t code[7] = {SEMZ_FILTER_X86, SEMZ_FILTER_POWERPC, SEMZ_FILTER_IA64, SEMZ_FILTER_ARM,
                                            SEMZ_FILTER_ARMTHUMB, SEMZ_FILTER_SPARC, SEMZ_FILTER_ARM64};
            h.filters |= (uint32_t)code[id - 4] << (4 * stage++);
        } else if (id == 0x21) {
            if (idx != nfilters) return fail(SEMZ_E_XZ_LZMA2_NOT_LAST, 0);
            if (pl != 1) return fail(SEMZ_E_XZ_LZMA2_PROPS, (uint32_t)pl);
            if (pp[0] >> 6) return fail(SEMZ_E_XZ_LZMA2_RESERVED, 0);
            const uint32_t o = pp[0] & 63;                // parse-lzma2-dict-size, lzma2.lisp:40-46
            if (o > 40) return fail(SEMZ_E_LZMA2_DICT_BYTE, o);
            h.dict_size = o == 40 ? 0xFFFFFFFFu : (2u | (o & 1)) << (11 + o / 2);
        } else {
            return fail(SEMZ_E_XZ_FILTER, (uint32_t)id);
        }
    }
    return h;
}

bool xz_blocks_from_index(const uint8_t *in, uint64_t in_len, std::vector<XzIndexedBlock> &blocks)
{
    std::vector<XzIndexedBlock> rev_streams;              // streams are found last to first
    std::vector<std::vector<XzIndexedBlock>> streams;
    uint64_t end = in_len;
    while (end > 0) {
        // stream padding: groups of four zero bytes (xz.lisp:118-134)
        while (end >= 4 && !in[end - 1] && !in[end - 2] && !in[end - 3] && !in[end - 4]) end -= 4;
        if (end == 0) break;
        if (end < 12 + 12) return false;
        const uint8_t *f = in + end - 12;
        if (f[10] != 0x59 || f[11] != 0x5A) return false;
        const uint32_t fcrc = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
        if (fcrc != crc32_host(f + 4, 6)) return false;
        const uint64_t bsize = 4ull * ((uint64_t)((uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24)) + 1);
        if (f[8] != 0 || (f[9] >> 4)) return false;
        const uint32_t check_type = f[9] & 15;
        if (end - 12 < bsize + 12) return false;
        const uint64_t idx_start = end - 12 - bsize;
        uint64_t p = idx_start;
        if (in[p++] != 0) return false;
        uint64_t count;
        if (xz_varint(in, &p, end - 12, &count) != SEMZ_OK) return false;
        if (count > (end - 12 - p) / 2) return false;
        std::vector<XzIndexedBlock> recs;
        recs.reserve((size_t)count);
        uint64_t total = 0;
        for (uint64_t k = 0; k < count; k++) {
            XzIndexedBlock b{};
            if (xz_varint(in, &p, end - 12, &b.unpadded) != SEMZ_OK) return false;
            if (xz_varint(in, &p, end - 12, &b.usize) != SEMZ_OK) return false;
            if (b.unpadded < 5 || b.unpadded > (1ull << 62)) return false;
            // an LZMA2 chunk makes at most 2 MiB out of its 11 bytes and more: a record that promises more than that
            // from the block's bytes is no prediction to allocate by (the forward walk reports the mismatch)
            if (b.usize > (1ull << 62) || (b.unpadded < (1ull << 40) && b.usize > (b.unpadded << 18) + (1u << 21))) return false;
            b.check_type = check_type;
            total += (b.unpadded + 3) & ~3ull;
            if (total > idx_start) return false;
            recs.push_back(b);
        }
        if (idx_start < total + 12) return false;
        uint64_t pos = idx_start - total;                 // first block header
        const uint64_t stream_start = pos - 12;
        for (auto &b : recs) {
            b.header_pos = pos;
            pos += (b.unpadded + 3) & ~3ull;
        }
        streams.push_back(std::move(recs));
        end = stream_start;
    }
    for (size_t s = streams.size(); s-- > 0;)
        for (auto &b : streams[s]) blocks.push_back(b);
    (void)rev_streams;
    return true;
}

}  // namespace semz

using namespace semz;

static semz_unit *alloc_units(size_t n)
{
    return (semz_unit *)calloc(n ? n : 1, sizeof(semz_unit));
}

extern "C" int semz_split(int fmt, const uint8_t *in, size_t in_len, const uint64_t *offsets, const uint64_t *out_sizes,
                          size_t noffsets, semz_unit **units_out, size_t *nunits_out, semz_stream_info *info)
{
    if (!units_out || !nunits_out || (!in && in_len)) return -1;
    *units_out = nullptr;
    *nunits_out = 0;
    semz_stream_info si;
    memset(&si, 0, sizeof si);
    auto finish = [&](semz_unit *u, size_t n) {
        *units_out = u;
        *nunits_out = n;
        if (info) *info = si;
        return 0;
    };
    if (fmt == SEMZ_FMT_DEFLATE || fmt == SEMZ_FMT_ZLIB || fmt == SEMZ_FMT_GZIP) {
        const uint32_t kind = fmt == SEMZ_FMT_GZIP ? SEMZ_UNIT_GZIP : fmt == SEMZ_FMT_ZLIB ? SEMZ_UNIT_ZLIB : SEMZ_UNIT_DEFLATE;
        std::vector<uint64_t> offs;
        size_t nindexed = 0;                              // leading members whose end is known without decoding
        if (noffsets) {
            if (!offsets) return -1;
            for (size_t i = 0; i < noffsets; i++) {
                if (offsets[i] > in_len || (i && offsets[i] < offsets[i - 1])) return -1;
                offs.push_back(offsets[i]);
            }
            nindexed = noffsets;
        } else if (fmt == SEMZ_FMT_GZIP) {
            // BGZF walk: every member names its own size in a "BC" extra subfield (gzip.lisp:10-29 parses the field)
            uint64_t pos = 0;
            while (pos < in_len) {
                GzipHeader h = parse_gzip_header(in, pos, in_len);
                offs.push_back(pos);
                if (h.status != SEMZ_OK || h.bgzf_size < 0 || (uint64_t)h.bgzf_size < h.header_len + 8 ||
                    pos + (uint64_t)h.bgzf_size > in_len)
                    break;                                // not indexable from here on: one sequential chain (semz_decode_stream)
                nindexed++;
                pos += (uint64_t)h.bgzf_size;
            }
            if (offs.empty()) offs.push_back(0);
        } else {
            offs.push_back(0);
        }
        const size_t n = offs.size();
        semz_unit *u = alloc_units(n);
        if (!u) return -5;
        uint64_t out_off = 0;
        bool sized = true;
        for (size_t i = 0; i < n; i++) {
            const uint64_t next = i + 1 < n ? offs[i + 1] : in_len;
            u[i].kind = kind;
            u[i].in_off = offs[i];
            // a gzip member may look ahead into its successor in all-members mode (interface.lisp:81-84)
            u[i].in_limit = fmt == SEMZ_FMT_GZIP ? in_len : next;
            uint64_t cap = 0;
            if (out_sizes) cap = out_sizes[i];
            else if (fmt == SEMZ_FMT_GZIP && i < nindexed && next >= offs[i] + 18)
                cap = (uint64_t)in[next - 4] | ((uint64_t)in[next - 3] << 8) | ((uint64_t)in[next - 2] << 16) | ((uint64_t)in[next - 1] << 24);
            else sized = false;
            u[i].out_off = out_off;
            u[i].out_cap = cap;
            out_off += cap;
        }
        si.reserved = (uint32_t)(out_sizes ? n : (fmt == SEMZ_FMT_GZIP ? nindexed : 0));
        (void)sized;
        si.total_out = out_off;                          // sum over the units whose size is known
        si.nmembers = (uint32_t)n;
        if (fmt == SEMZ_FMT_GZIP && in_len) {
            GzipHeader h = parse_gzip_header(in, 0, in_len);
            si.status = h.status;
            si.detail = h.detail;
            // textp | extra flags << 8 | operating system << 16; mtime; extra field span; name span; comment span
            si.first_header[0] = h.textp | (h.xfl << 8) | (h.os << 16); si.first_header[1] = h.mtime;
            si.first_header[2] = (uint32_t)h.extra_off; si.first_header[3] = (uint32_t)h.extra_len;
            si.first_header[4] = (uint32_t)h.name_off; si.first_header[5] = (uint32_t)h.name_len;
            si.first_header[6] = (uint32_t)h.comment_off; si.first_header[7] = (uint32_t)h.comment_len;
        } else if (fmt == SEMZ_FMT_ZLIB && in_len >= 2) {
            si.first_header[0] = 1u << ((in[0] >> 4) + 8); si.first_header[1] = in[1] >> 6; si.first_header[2] = (in[1] >> 5) & 1;
            if (((in[1] >> 5) & 1) && in_len >= 6)
                si.first_header[3] = ((uint32_t)in[2] << 24) | ((uint32_t)in[3] << 16) | ((uint32_t)in[4] << 8) | in[5];
        }
        return finish(u, n);
    }
    if (fmt == SEMZ_FMT_BZIP2) {
        uint32_t level = 0, det = 0;
        si.status = parse_bzip2_header(in, 0, in_len, &level, &det);
        si.detail = det;
        si.first_header[0] = level;
        std::vector<BzCandidate> c = scan_bzip2(in, in_len);
        size_t n = 0;
        for (auto &x : c) n += !x.footer;
        semz_unit *u = alloc_units(n);
        if (!u) return -5;
        size_t k = 0;
        for (auto &x : c) {
            if (x.footer) { si.nmembers++; continue; }
            u[k].kind = SEMZ_UNIT_BZIP2_BLOCK;
            u[k].in_off = x.bit >> 3;
            u[k].in_bit_off = (uint32_t)(x.bit & 7);
            u[k].in_limit = in_len;
            u[k].out_off = SEMZ_OUT_AUTO;
            u[k].param0 = x.level;
            k++;
        }
        return finish(u, n);
    }
    if (fmt == SEMZ_FMT_XZ) {
        uint32_t ct = 0, det = 0;
        uint64_t hend = 0;
        si.status = parse_xz_stream_header(in, 0, in_len, false, &ct, &hend, &det);
        si.detail = det;
        si.first_header[0] = ct;
        std::vector<XzIndexedBlock> blocks;
        if (si.status != SEMZ_OK || !xz_blocks_from_index(in, in_len, blocks)) return finish(alloc_units(0), 0);
        semz_unit *u = alloc_units(blocks.size());
        if (!u) return -5;
        uint64_t out_off = 0;
        size_t n = 0;
        for (auto &b : blocks) {
            if (b.header_pos >= in_len || in[b.header_pos] == 0) break;
            XzBlockHeader h = parse_xz_block_header(in, b.header_pos, in_len);
            if (h.status != SEMZ_OK) break;               // the forward pass of semz_decode_stream reports it
            semz_unit &x = u[n++];
            x.kind = SEMZ_UNIT_XZ_BLOCK;
            x.in_off = b.header_pos + h.header_size + 4;
            x.in_limit = in_len;
            x.out_off = out_off;
            x.out_cap = b.usize;
            x.flags = (h.have_csize ? SEMZ_UF_HAVE_IN_SIZE : 0) | (h.have_usize ? SEMZ_UF_HAVE_OUT_SIZE : 0);
            x.size_in = h.csize;
            x.size_out = h.usize;
            x.param0 = h.dict_size;
            x.param1 = b.check_type;
            x.param2 = h.bcj_offset;
            x.param3 = h.delta_dists;
            x.filters = h.filters;
            uint64_t next_off;
            if (__builtin_add_overflow(out_off, b.usize, &next_off)) { n--; break; }   // (cannot happen with the bound above)
            out_off = next_off;
        }
        si.total_out = out_off;
        si.nmembers = 1;
        return finish(u, n);
    }
    return -1;
}
