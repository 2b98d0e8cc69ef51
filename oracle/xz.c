/*
 * xz.c -- oracle restatement of xz.lisp and xz-bcj-filters.lisp plus
 * the delta filter.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   stream header                 xz.lisp:78-112
 *   multibyte integers            xz.lisp:141-162
 *   block header + filter chain   xz.lisp:260-379, 389-407
 *   block end (sizes, padding, check)  xz.lisp:426-457
 *   index + footer                xz.lisp:166-233
 *   padding / next member         xz.lisp:118-134
 *   delta filter                  xz.lisp:238-258
 *   x86 BCJ                       xz-bcj-filters.lisp:169-219 (+ wrapper 14-43)
 *   ARM, ARM64, ARM-Thumb, IA-64, PowerPC, SPARC BCJ   xz-bcj-filters.lisp:45-167
 *
 * The reference streams filters over 8 KiB buffers; both filters here are
 * applied once over the whole block, which gives the same bytes: the delta
 * filter carries its whole state (256-byte history, pos) across buffers, and
 * the x86 filter's per-call clamp of prev-pos (xz-bcj-filters.lisp:172-173)
 * only replaces "offset > 5 -> prev-mask = 0" by five shifts that clear
 * prev-mask as well, while bytes it cannot process yet are carried over.
 */
#include "lzma_src.h"

/* ---- x86 BCJ inverse (decoder direction), whole buffer ---- */
void orc_bcj_x86(uint8_t *buf, size_t n, uint32_t start_offset)
{
    static const uint8_t allowed[8] = {1, 1, 1, 0, 1, 0, 0, 0};
    static const uint8_t bitnum[8] = {0, 1, 2, 2, 3, 3, 3, 3};
    uint32_t now_pos = start_offset, prev_mask = 0, prev_pos = (uint32_t)-5;
    if (n < 5) return;
    if ((uint32_t)(now_pos - prev_pos) > 5) prev_pos = now_pos - 5;
    size_t limit = n - 5, i = 0;
    while (i <= limit) {
        if (buf[i] != 0xE8 && buf[i] != 0xE9) { i++; continue; }
        uint32_t offset = now_pos + (uint32_t)i - prev_pos;
        prev_pos = now_pos + (uint32_t)i;
        if (offset > 5) prev_mask = 0;
        else for (uint32_t k = 0; k < offset; k++) prev_mask = (prev_mask & 0x77) << 1;
        uint8_t b = buf[i + 4];
        if ((b == 0x00 || b == 0xFF) && allowed[(prev_mask >> 1) & 7] && (prev_mask >> 1) < 0x10) {
            uint32_t src = (uint32_t)buf[i + 1] | ((uint32_t)buf[i + 2] << 8) | ((uint32_t)buf[i + 3] << 16) |
                           ((uint32_t)buf[i + 4] << 24);
            uint32_t dest;
            for (;;) {
                dest = src - (now_pos + (uint32_t)i + 5);
                if (prev_mask == 0) break;
                unsigned j = bitnum[prev_mask >> 1];
                b = (uint8_t)(dest >> (24 - j * 8));
                if (!(b == 0x00 || b == 0xFF)) break;
                src = dest ^ ((1u << (32 - j * 8)) - 1);
            }
            buf[i + 4] = (uint8_t)~(((dest >> 24) & 1) - 1);
            buf[i + 1] = (uint8_t)dest;
            buf[i + 2] = (uint8_t)(dest >> 8);
            buf[i + 3] = (uint8_t)(dest >> 16);
            i += 5;
            prev_mask = 0;
        } else {
            i++;
            prev_mask |= 1;
            if (b == 0x00 || b == 0xFF) prev_mask |= 0x10;
        }
    }
}

/* ---- the other BCJ inverses (xz-bcj-filters.lisp:45-167), whole buffer ----
 * id: xz filter id 0x05 PowerPC, 0x06 IA-64, 0x07 ARM, 0x08 ARM-Thumb, 0x09 SPARC, 0x0A ARM64.
 * The reference runs them over 8 KiB buffers and carries the bytes it cannot process yet
 * over to the next call, with i counted from the start of the data: one pass over the whole
 * block visits the same positions with the same now-pos + i. */
static uint32_t ld24le(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
static uint32_t ld32le(const uint8_t *p) { return ld24le(p) | ((uint32_t)p[3] << 24); }
static void st32le(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

void orc_bcj_simple(uint8_t *buf, size_t n, uint32_t now_pos, unsigned id)
{
    size_t i = 0;
    if (id == 0x07) {                                               /* arm, :45-53 */
        for (; i + 4 <= n; i += 4)
            if (buf[i + 3] == 0xEB) {
                uint32_t v = ((ld24le(buf + i) << 2) - (now_pos + (uint32_t)i + 8)) >> 2;
                buf[i] = (uint8_t)v; buf[i + 1] = (uint8_t)(v >> 8); buf[i + 2] = (uint8_t)(v >> 16);
            }
    } else if (id == 0x0A) {                                        /* arm64, :55-80 */
        for (; i + 4 <= n; i += 4) {
            uint32_t pc = now_pos + (uint32_t)i, instr = ld32le(buf + i);
            if ((instr >> 26) == 0x25) {
                st32le(buf + i, ((instr - (pc >> 2)) & 0x03FFFFFFu) | 0x94000000u);
            } else if ((instr & 0x9F000000u) == 0x90000000u) {
                uint32_t src = ((instr >> 29) & 3) | ((instr >> 3) & 0x001FFFFCu);
                if (((src + 0x00020000u) & 0x001C0000u) == 0) {
                    uint32_t dest = src - (pc >> 12);
                    st32le(buf + i, (instr & 0x9000001Fu) | ((dest & 3) << 29) | ((dest & 0x0003FFFCu) << 3) |
                                    ((0u - (dest & 0x00020000u)) & 0x00E00000u));
                }
            }
        }
    } else if (id == 0x08) {                                        /* armthumb, :82-99 */
        for (; i + 4 <= n; i += 2)
            if ((buf[i + 1] & 0xF8) == 0xF0 && (buf[i + 3] & 0xF8) == 0xF8) {
                uint32_t src = ((((uint32_t)buf[i + 1] & 7) << 19) | ((uint32_t)buf[i] << 11) |
                                (((uint32_t)buf[i + 3] & 7) << 8) | (uint32_t)buf[i + 2]) << 1;
                uint32_t dest = (src - (now_pos + (uint32_t)i + 4)) >> 1;
                buf[i + 1] = (uint8_t)(0xF0 | ((dest >> 19) & 7));
                buf[i] = (uint8_t)(dest >> 11);
                buf[i + 3] = (uint8_t)(0xF8 | ((dest >> 8) & 7));
                buf[i + 2] = (uint8_t)dest;
                i += 2;
            }
    } else if (id == 0x06) {                                        /* ia64, :101-130 */
        static const uint8_t masks[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 6, 6, 0, 0, 7, 7, 4, 4, 0, 0, 4, 4, 0, 0};
        for (; i + 16 <= n; i += 16) {
            unsigned mask = masks[buf[i] & 0x1F];
            for (unsigned slot = 0, bit_pos = 5; slot < 3; slot++, bit_pos += 41) {
                if (!((mask >> slot) & 1)) continue;
                size_t byte_pos = bit_pos >> 3;
                unsigned bit_res = bit_pos & 7;
                uint64_t instruction = 0;
                for (int k = 0; k < 6; k++) instruction |= (uint64_t)buf[i + byte_pos + k] << (8 * k);
                uint64_t inst_norm = instruction >> bit_res;
                if (((inst_norm >> 37) & 0xF) == 0x5 && ((inst_norm >> 9) & 0x7) == 0) {
                    uint32_t src = (uint32_t)(((inst_norm >> 13) & 0xFFFFF) | (((inst_norm >> 36) & 1) << 20)) << 4;
                    uint32_t dest = (src - (now_pos + (uint32_t)i)) >> 4;
                    uint64_t v = (((uint64_t)1 << bit_res) - 1) & instruction;
                    v |= ((((uint64_t)(dest & 0x100000)) << (36 - 20)) | ((uint64_t)(dest & 0xFFFFF) << 13) |
                          (inst_norm & ~((uint64_t)0x8FFFFF << 13))) << bit_res;
                    for (int k = 0; k < 6; k++) buf[i + byte_pos + k] = (uint8_t)(v >> (8 * k));
                }
            }
        }
    } else if (id == 0x05) {                                        /* powerpc, :132-148 */
        for (; i + 4 <= n; i += 4)
            if ((buf[i] >> 2) == 0x12 && (buf[i + 3] & 3) == 1) {
                uint32_t src = (((uint32_t)buf[i] & 3) << 24) | ((uint32_t)buf[i + 1] << 16) | ((uint32_t)buf[i + 2] << 8) |
                               ((uint32_t)buf[i + 3] & ~3u);
                uint32_t dest = src - (now_pos + (uint32_t)i);
                buf[i] = (uint8_t)(0x48 | ((dest >> 24) & 3));
                buf[i + 1] = (uint8_t)(dest >> 16);
                buf[i + 2] = (uint8_t)(dest >> 8);
                buf[i + 3] = (uint8_t)((buf[i + 3] & 3) | (dest & 0xFF));
            }
    } else if (id == 0x09) {                                        /* sparc, :150-167 */
        for (; i + 4 <= n; i += 4)
            if ((buf[i] == 0x40 && (buf[i + 1] & 0xC0) == 0x00) || (buf[i] == 0x7F && (buf[i + 1] & 0xC0) == 0xC0)) {
                uint32_t src = ((uint32_t)buf[i] << 24) | ((uint32_t)buf[i + 1] << 16) | ((uint32_t)buf[i + 2] << 8) | buf[i + 3];
                uint32_t dest = ((src << 2) - (now_pos + (uint32_t)i)) >> 2;
                uint32_t v = (((0u - ((dest >> 22) & 1)) << 22) & 0x3FFFFFFFu) | (dest & 0x3FFFFFu) | 0x40000000u;
                buf[i] = (uint8_t)(v >> 24); buf[i + 1] = (uint8_t)(v >> 16); buf[i + 2] = (uint8_t)(v >> 8); buf[i + 3] = (uint8_t)v;
            }
    }
}

/* ---- delta filter inverse (xz.lisp:238-258) ---- */
void orc_delta(uint8_t *buf, size_t n, unsigned distance)
{
    uint8_t delta[256];
    unsigned pos = 0;
    memset(delta, 0, sizeof delta);
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp = (uint8_t)(delta[(distance + pos) & 0xFF] + buf[i]);
        delta[pos] = tmp;
        buf[i] = tmp;
        pos = (pos - 1) & 0xFF;
    }
}

/* ---- multibyte integers (xz.lisp:141-154) ---- */
static uint64_t xz_varint(osrc *s)
{
    unsigned x0 = osrc_byte(s);
    if (x0 < 0x80) return x0;
    uint64_t value = x0 & 0x7F;
    for (int i = 7; i < 63; i += 7) {
        unsigned byte = osrc_byte(s);
        value |= (uint64_t)(byte & 0x7F) << i;
        if (byte == 0) orc_die(s->ctx, SEMZ_E_XZ_VARINT_ZERO, 0);
        if (!(byte & 0x80)) return value;
    }
    orc_die(s->ctx, SEMZ_E_XZ_VARINT_LONG, 0);
    return 0;
}

static size_t push_varint(uint64_t v, uint8_t *dst)     /* xz.lisp:156-162 */
{
    size_t n = 0;
    for (;;) {
        dst[n++] = (uint8_t)(v & 0x7F);
        v >>= 7;
        if (!v) return n;
        dst[n - 1] |= 0x80;
    }
}

static const uint8_t check_sizes[16] = {0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64};

static void compute_check(unsigned type, const uint8_t *p, size_t n, uint8_t out[32])
{
    memset(out, 0, 32);
    if (type == 1) { uint32_t c = orc_crc32(p, n); memcpy(out, &c, 4); }
    else if (type == 4) { uint64_t c = orc_crc64(p, n); memcpy(out, &c, 8); }
    else if (type == 10) orc_sha256(p, n, out);
}

typedef struct xzblk { uint64_t unpadded, usize; } xzblk;

/* handle-xz-block + :data + :block-end for one block; s is positioned after the header CRC */
static void xz_block(orc_ctx *c, osrc *s, const uint8_t *hdr, size_t hlen, unsigned check_type,
                     uint8_t *out, size_t out_cap, uint64_t *pos, xzblk *blk, uint8_t check_out[32])
{
    unsigned flags = hdr[1], nfilters = 1 + (flags & 3);
    if ((flags >> 2) & 15) orc_die(c, SEMZ_E_XZ_BLOCK_RESERVED, 0);
    int have_csize = (flags >> 6) & 1, have_usize = (flags >> 7) & 1;
    uint64_t csize = 0, usize = 0, fid[4];
    const uint8_t *fprops[4];
    uint64_t fplen[4];
    /* header fields: an eof inside the header is "Block header is truncated." (xz.lisp:273-288) */
    orc_ctx hc;
    osrc hs;
    osrc_init(&hs, &hc, hdr, hlen, 2);
    if (setjmp(hc.jb)) orc_die(c, hc.status == SEMZ_EOF ? SEMZ_E_XZ_BLOCK_TRUNCATED : hc.status, hc.detail);
    if (have_csize) csize = xz_varint(&hs);
    if (have_usize) usize = xz_varint(&hs);
    for (unsigned f = 0; f < nfilters; f++) {
        fid[f] = xz_varint(&hs);
        fplen[f] = xz_varint(&hs);
        if (fplen[f] > osrc_left(&hs)) orc_die(&hc, SEMZ_EOF, 0);
        fprops[f] = hdr + hs.pos;
        hs.pos += fplen[f];
    }
    for (uint64_t i = hs.pos; i < hlen; i++)
        if (hdr[i]) orc_die(c, SEMZ_E_XZ_BLOCK_HDR_PADDING, 0);

    /* filter chain, last filter first (xz.lisp:311-377) */
    uint32_t dict_size = 0, bcj_off = 0;
    for (unsigned idx = nfilters; idx >= 1; idx--) {
        uint64_t id = fid[idx - 1], pl = fplen[idx - 1];
        const uint8_t *pp = fprops[idx - 1];
        if (id == 0x03) {
            if (idx == nfilters) orc_die(c, SEMZ_E_XZ_DELTA_LAST, 0);
            if (pl != 1) orc_die(c, SEMZ_E_XZ_DELTA_PROPS, (uint32_t)pl);
        } else if (id >= 0x04 && id <= 0x0A) {
            static const uint8_t align[7] = {1, 4, 16, 4, 2, 4, 4};
            if (idx == nfilters) orc_die(c, SEMZ_E_XZ_BCJ_LAST, 0);
            if (pl == 0) bcj_off = 0;
            else if (pl == 4) bcj_off = (uint32_t)pp[0] | ((uint32_t)pp[1] << 8) | ((uint32_t)pp[2] << 16) | ((uint32_t)pp[3] << 24);
            else orc_die(c, SEMZ_E_XZ_BCJ_PROPS, (uint32_t)pl);
            if (bcj_off % align[id - 4]) orc_die(c, SEMZ_E_XZ_BCJ_ALIGN, bcj_off);
        } else if (id == 0x21) {
            if (idx != nfilters) orc_die(c, SEMZ_E_XZ_LZMA2_NOT_LAST, 0);
            if (pl != 1) orc_die(c, SEMZ_E_XZ_LZMA2_PROPS, (uint32_t)pl);
            if (pp[0] >> 6) orc_die(c, SEMZ_E_XZ_LZMA2_RESERVED, 0);
            unsigned o = pp[0] & 63;                                  /* parse-lzma2-dict-size, lzma2.lisp:40-46 */
            if (o > 40) orc_die(c, SEMZ_E_LZMA2_DICT_BYTE, o);
            dict_size = o == 40 ? 0xFFFFFFFFu : (2u | (o & 1)) << (11 + o / 2);
        } else {
            orc_die(c, SEMZ_E_XZ_FILTER, (uint32_t)id);
        }
    }

    /* :data */
    uint64_t start = *pos, data_start = s->pos;
    osrc_rebind(s, c);
    if (have_csize) osrc_push_limit(s, csize, SEMZ_E_XZ_BLOCK_OVERRUN);
    orc_lzma2_run(c, s, dict_size, out, out_cap, pos);
    uint64_t real_csize = s->pos - data_start;
    for (int idx = (int)nfilters - 1; idx >= 1; idx--) {              /* innermost filter first */
        if (fid[idx - 1] == 0x03) orc_delta(out + start, *pos - start, 1 + fprops[idx - 1][0]);
        else {
            uint32_t off = fplen[idx - 1] == 4 ? ((uint32_t)fprops[idx - 1][0] | ((uint32_t)fprops[idx - 1][1] << 8) |
                                                  ((uint32_t)fprops[idx - 1][2] << 16) | ((uint32_t)fprops[idx - 1][3] << 24)) : 0;
            if (fid[idx - 1] == 0x04) orc_bcj_x86(out + start, *pos - start, off);
            else orc_bcj_simple(out + start, *pos - start, off, (unsigned)fid[idx - 1]);
        }
    }
    /* :block-end (xz.lisp:426-457) */
    if (have_csize && osrc_pop_limit(s) != 0) orc_die(c, SEMZ_E_XZ_BLOCK_SHORT, 0);
    if (have_usize && *pos - start != usize) orc_die(c, SEMZ_E_XZ_BLOCK_OUT_SIZE, 0);
    for (uint64_t k = 0; k < ((4 - (real_csize & 3)) & 3); k++)
        if (osrc_byte(s)) orc_die(c, SEMZ_E_XZ_BLOCK_PADDING, 0);
    unsigned cs = check_sizes[check_type];
    uint8_t want[64], have[32];
    osrc_read(s, want, cs);
    compute_check(check_type, out + start, *pos - start, have);
    memcpy(check_out, have, 32);
    blk->unpadded = 4 + hlen + real_csize + cs;
    blk->usize = *pos - start;
    if (memcmp(want, have, cs)) {
        uint32_t w0 = 0;
        memcpy(&w0, want, cs < 4 ? cs : 4);
        orc_die(c, SEMZ_E_XZ_BLOCK_CHECK, w0);
    }
}

/* one member; skip_magic: the 6 magic bytes were already consumed (xz.lisp:88-91) */
static void xz_member(orc_ctx *c, osrc *s, int skip_magic, uint8_t *out, size_t out_cap, uint64_t *pos,
                      orc_result *res)
{
    if (!skip_magic && osrc_le(s, 6) != 0x005A587A37FDull) orc_die(c, SEMZ_E_XZ_MAGIC, 0);
    if (osrc_byte(s) != 0) orc_die(c, SEMZ_E_XZ_RESERVED_BYTE, 0);
    unsigned header2 = osrc_byte(s), type = header2 & 15;
    uint32_t hcrc = (uint32_t)osrc_le(s, 4);
    uint8_t hb[2] = {0, (uint8_t)header2};
    if (hcrc != orc_crc32(hb, 2)) orc_die(c, SEMZ_E_XZ_HEADER_CRC, 0);
    if (header2 >> 4) orc_die(c, SEMZ_E_XZ_RESERVED_FLAGS, 0);
    if (!(type == 0 || type == 1 || type == 4 || type == 10)) orc_die(c, SEMZ_E_XZ_CHECK_TYPE, type);
    res->detail = type;
    unsigned stream_flags = header2 << 8;

    size_t nblk = 0, capblk = 16;
    xzblk *blks = (xzblk *)malloc(capblk * sizeof(xzblk));
    uint8_t *hdr = (uint8_t *)malloc(1024), *ridx = NULL;
    orc_ctx inner;
    osrc_rebind(s, &inner);
    if (setjmp(inner.jb)) {
        free(blks); free(hdr); free(ridx);
        osrc_rebind(s, c);
        orc_die(c, inner.status, inner.detail);
    }
    for (;;) {
        unsigned first = osrc_byte(s);
        if (first == 0) break;
        size_t hlen = 4 * (size_t)first;
        hdr[0] = (uint8_t)first;
        osrc_read(s, hdr + 1, hlen - 1);
        uint32_t crc = (uint32_t)osrc_le(s, 4);
        if (crc != orc_crc32(hdr, hlen)) orc_die(&inner, SEMZ_E_XZ_BLOCK_HDR_CRC, crc);
        if (nblk == capblk) { capblk *= 2; blks = (xzblk *)realloc(blks, capblk * sizeof(xzblk)); }
        /* the block counts as observed as soon as its header is accepted (xz.lisp:294-295) */
        xz_block(&inner, s, hdr, hlen, type, out, out_cap, pos, &blks[nblk], res->check);
        osrc_rebind(s, &inner);
        nblk++;
        res->check_len = check_sizes[type];
    }
    /* finish-xz-member (xz.lisp:166-233) */
    size_t rcap = 16 + 20 * (nblk + 2), rn = 0;
    ridx = (uint8_t *)malloc(rcap);
    ridx[rn++] = 0;
    uint64_t count = xz_varint(s);
    rn += push_varint(count, ridx + rn);
    size_t mism_count = count != nblk, mism_sizes = 0;
    for (uint64_t k = 0; k < count; k++) {
        uint64_t a = xz_varint(s), b = xz_varint(s);
        if (rn + 24 > rcap) { rcap *= 2; ridx = (uint8_t *)realloc(ridx, rcap); }
        rn += push_varint(a, ridx + rn);
        rn += push_varint(b, ridx + rn);
        if (k < nblk && (a != blks[k].unpadded || b != blks[k].usize)) mism_sizes = 1;
    }
    while (rn & 3) {
        if (osrc_byte(s)) orc_die(&inner, SEMZ_E_XZ_INDEX_PADDING, 0);
        ridx[rn++] = 0;
    }
    if (mism_count) orc_die(&inner, SEMZ_E_XZ_INDEX_COUNT, 0);
    if (mism_sizes) orc_die(&inner, SEMZ_E_XZ_INDEX_SIZES, 0);
    uint32_t icrc = (uint32_t)osrc_le(s, 4);
    if (icrc != orc_crc32(ridx, rn)) orc_die(&inner, SEMZ_E_XZ_INDEX_CRC, icrc);
    uint8_t footer[12];
    osrc_read(s, footer, 12);
    uint32_t fcrc, bsize;
    memcpy(&fcrc, footer, 4);
    memcpy(&bsize, footer + 4, 4);
    if (fcrc != orc_crc32(footer + 4, 6)) orc_die(&inner, SEMZ_E_XZ_FOOTER_CRC, fcrc);
    if (4ull * ((uint64_t)bsize + 1) != 4 + rn) orc_die(&inner, SEMZ_E_XZ_BACKWARD_SIZE, 0);
    if (stream_flags != ((unsigned)footer[8] | ((unsigned)footer[9] << 8))) orc_die(&inner, SEMZ_E_XZ_FOOTER_FLAGS, 0);
    if (footer[10] != 0x59 || footer[11] != 0x5A) orc_die(&inner, SEMZ_E_XZ_FOOTER_MAGIC, 0);
    free(blks); free(hdr); free(ridx);
    osrc_rebind(s, c);
}

void orc_xz_member_ex(const uint8_t *in, size_t in_len, int skip_magic, uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    osrc s;
    uint64_t pos = 0;
    memset(res, 0, sizeof *res);
    osrc_init(&s, &c, in, in_len, 0);
    if (setjmp(c.jb)) {
        res->status = c.status;
        if (c.status != SEMZ_OK) res->detail = c.detail;
        res->in_end_bit = 8 * s.pos;
        res->out_len = pos;
        return;
    }
    xz_member(&c, &s, skip_magic, out, out_cap, &pos, res);
    res->status = SEMZ_OK;
    res->in_end_bit = 8 * s.pos;
    res->out_len = pos;
}

void orc_xz_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_xz_member_ex(in, in_len, 0, out, out_cap, res);
}

/*
 * make-reset-state for xz (xz.lisp:118-134): skips 4-byte groups of zero
 * padding after a member at `pos`.  *next = offset where the next member's
 * header continues AFTER its 6 magic bytes (skip-magic), or in_len at a clean
 * end; *status = SEMZ_OK / SEMZ_E_XZ_TRAILING_GARBAGE / SEMZ_E_XZ_PADDING / SEMZ_EOF.
 */
void orc_xz_next_member_start(const uint8_t *in, size_t in_len, size_t pos, size_t *next, int *status)
{
    *status = SEMZ_OK;
    for (;;) {
        size_t left = in_len - pos;
        if (left == 0) { *next = in_len; return; }
        if (left < 4) { *status = SEMZ_E_XZ_PADDING; *next = pos; return; }
        const uint8_t *b = in + pos;
        pos += 4;
        if (b[0] | b[1] | b[2] | b[3]) {
            if (b[0] == 0xFD && b[1] == 0x37 && b[2] == 0x7A && b[3] == 0x58) {
                if (pos >= in_len) { *status = SEMZ_EOF; *next = pos; return; }
                if (in[pos++] == 0x5A) {
                    if (pos >= in_len) { *status = SEMZ_EOF; *next = pos; return; }
                    if (in[pos++] == 0x00) { *next = pos; return; }
                }
            }
            *status = SEMZ_E_XZ_TRAILING_GARBAGE;
            *next = pos;
            return;
        }
    }
}
