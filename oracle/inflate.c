/*
 * inflate.c -- oracle restatement of deflate.lisp, zlib.lisp, gzip.lisp.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference decodes in chunks of <= 32 KiB + 257 bytes (deflate.lisp:416,
 * 265-268); chunking has no observable effect, so this restatement writes the
 * whole member into one output span.  `pointer` of deflate.lisp:278 (position
 * in the window+result buffer) corresponds to prefix_amount + produced here:
 * after every flush-dict-buffer the reference's pointer is >= window-size, so
 * min(pointer, wsize) == min(prefix_amount + produced, wsize) at all times.
 */
#include "bitio.h"

/* deflate.lisp:14-34 */
static const uint16_t LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DIST_BASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t DIST_EXTRA[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
/* deflate.lisp:174 */
static const uint8_t CLEN_ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

#define MIN_BLOCK_BITS 10  /* +min-deflate-block-bitsize+, deflate.lisp:4-12 */

enum { MODE_NORMAL = 0, MODE_LITERALS_ONLY = 1, MODE_SINGLE_CODE = 2 };

typedef struct dhi {
    orc_ht litlen, dist, clen;
    int special;
} dhi;

/* lengths->dist-ht, deflate.lisp:91-122 */
static void build_dist(orc_ctx *c, dhi *d, const uint8_t *lengths, int n, long overread)
{
    if (n == 1 && lengths[0] == 0) {
        uint8_t illegal[32] = {0};
        illegal[30] = illegal[31] = 1;              /* deflate.lisp:63-69 */
        ht_build(c, &d->dist, illegal, 32, overread, 1);
        d->special = MODE_LITERALS_ONLY;
        return;
    }
    int one = -1, singular = 1;                     /* find-singular-one, deflate.lisp:73-84 */
    for (int i = 0; i < n && singular; i++) {
        if (lengths[i] == 0) continue;
        if (one >= 0 || lengths[i] != 1) singular = 0;
        else one = i;
    }
    if (singular && one >= 0) {
        uint8_t a[32] = {0};
        a[one] = 1;
        a[one == 31 ? 30 : 31] = 1;
        ht_build(c, &d->dist, a, 32, overread, 1);
        d->special = MODE_SINGLE_CODE;
        return;
    }
    ht_build(c, &d->dist, lengths, n, overread, 1);
    d->special = MODE_NORMAL;
}

/* lengths->dhi, deflate.lisp:126-136 */
static void build_dhi(orc_ctx *c, dhi *d, const uint8_t *lengths, int nlit, int ndist, long overread)
{
    ht_build(c, &d->litlen, lengths, nlit, overread, 1);
    build_dist(c, d, lengths + nlit, ndist, overread);
}

/* fixed trees, deflate.lisp:146-158 */
static void build_fixed(orc_ctx *c, dhi *d, long overread)
{
    uint8_t l[288 + 32];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    for (int i = 0; i < 32; i++) l[288 + i] = 5;
    build_dhi(c, d, l, 288, 32, overread);
}

/* read-dynamic-dhi, deflate.lisp:160-201 */
static void read_dynamic(orc_ctx *c, bitrd *b, dhi *d, long overread)
{
    int nlit = 257 + (int)lbr_read(b, 5);
    int ndist = 1 + (int)lbr_read(b, 5);
    int nclen = 4 + (int)lbr_read(b, 4);
    int total = nlit + ndist;
    uint8_t cl[19] = {0};
    uint8_t lens[257 + 31 + 1 + 31];
    for (int i = 0; i < nclen; i++) cl[CLEN_ORDER[i]] = (uint8_t)lbr_read(b, 3);
    ht_build(c, &d->clen, cl, 19, overread, 1);
    int i = 0;
    while (i < total) {
        unsigned code = ht_read(b, &d->clen, 1);
        if (code >= 16) {
            int n = code == 16 ? 3 + (int)lbr_read(b, 2)
                  : code == 17 ? 3 + (int)lbr_read(b, 3)
                               : 11 + (int)lbr_read(b, 7);
            if (i + n > total) orc_die(c, SEMZ_E_DEFLATE_CODELEN_BOUNDS, 0);
            uint8_t v = 0;
            if (code == 16) {
                if (i == 0) orc_die(c, SEMZ_E_DEFLATE_REPEAT_FIRST, 0);
                v = lens[i - 1];
            }
            memset(lens + i, v, n);
            i += n;
        } else {
            lens[i++] = (uint8_t)code;
        }
    }
    build_dhi(c, d, lens, nlit, ndist, overread);
}

typedef struct outbuf {
    uint8_t *out;
    size_t cap, n;
    const uint8_t *prefix;   /* last prefix_amount bytes of the preset history */
    size_t prefix_amount;
} outbuf;

/* decode-huffman-data, deflate.lisp:272-338 (without the chunk threshold) */
static void decode_block(orc_ctx *c, bitrd *b, const dhi *d, outbuf *o, uint32_t wsize)
{
    for (;;) {
        unsigned code = ht_read(b, &d->litlen, 1);
        if (code <= 255) {
            if (o->n >= o->cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
            o->out[o->n++] = (uint8_t)code;
        } else if (code == 256) {
            return;
        } else if (code <= 285) {
            if (d->special == MODE_LITERALS_ONLY) orc_die(c, SEMZ_E_DEFLATE_LEN_IN_LITONLY, code);
            unsigned len = LEN_BASE[code - 257] + (unsigned)lbr_read(b, LEN_EXTRA[code - 257]);
            unsigned dc = ht_read(b, &d->dist, 1);
            if (dc > 29)
                orc_die(c, d->special == MODE_SINGLE_CODE ? SEMZ_E_DEFLATE_SINGLE_DIST
                                                          : SEMZ_E_DEFLATE_DIST_CODE, dc);
            unsigned dist = DIST_BASE[dc] + (unsigned)lbr_read(b, DIST_EXTRA[dc]);
            uint64_t pointer = o->prefix_amount + o->n;
            uint64_t lim = pointer < wsize ? pointer : wsize;
            if (dist > lim) orc_die(c, SEMZ_E_DEFLATE_DISTANCE, dist);
            if (o->n + len > o->cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
            for (unsigned k = 0; k < len; k++) {         /* copy-match, common.lisp:324-333 */
                uint64_t src = pointer + k - dist;       /* index in prefix||out */
                o->out[o->n + k] = src < o->prefix_amount ? o->prefix[src]
                                                          : o->out[src - o->prefix_amount];
            }
            o->n += len;
        } else {
            orc_die(c, SEMZ_E_DEFLATE_LITLEN_CODE, code);
        }
    }
}

static void free_dhi(dhi *d)
{
    free(d->litlen.table); free(d->dist.table); free(d->clen.table);
    memset(d, 0, sizeof *d);
}

/* next-deflate-chunk block state machine, deflate.lisp:365-410 */
static void inflate_body(orc_ctx *c, bitrd *b, unsigned trailing_bits, uint32_t wsize,
                         outbuf *o, dhi *dyn, dhi *fix_final, dhi *fix_nonfinal)
{
    int final;
    do {
        final = (int)lbr_read(b, 1);
        unsigned type = (unsigned)lbr_read(b, 2);
        switch (type) {
        case 0: {
            br_flush_byte(b);
            unsigned len = (unsigned)lbr_read(b, 16);
            unsigned chk = (unsigned)lbr_read(b, 16);
            if (chk != (len ^ 0xFFFFu)) orc_die(c, SEMZ_E_DEFLATE_STORED_LEN, chk);
            if (o->n + len > o->cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
            if (b->nbits - b->pos < 8ull * len) orc_die(c, SEMZ_EOF, 0);   /* decode-uncompressed-data :350 */
            memcpy(o->out + o->n, b->in + (b->pos >> 3), len);
            o->n += len;
            b->pos += 8ull * len;
            break;
        }
        case 1: {
            dhi *f = final ? fix_final : fix_nonfinal;   /* deflate.lisp:402-404 */
            if (!f->litlen.table) build_fixed(c, f, final ? 0 : MIN_BLOCK_BITS);
            decode_block(c, b, f, o, wsize);
            break;
        }
        case 2:
            read_dynamic(c, b, dyn, (final ? 0 : MIN_BLOCK_BITS) + (long)trailing_bits);
            decode_block(c, b, dyn, o, wsize);
            break;
        default:
            orc_die(c, SEMZ_E_DEFLATE_BTYPE, 0);
        }
    } while (!final);
}

typedef struct inflate_scratch { dhi dyn, ff, fn; } inflate_scratch;

static void inflate_run(orc_ctx *c, bitrd *b, unsigned trailing_bits, uint32_t wsize,
                        const uint8_t *prefix, size_t prefix_len, outbuf *o, inflate_scratch *s)
{
    /* :deflate ctor, deflate.lisp:412-430: keep the last min(wsize, len) prefix bytes */
    size_t amount = prefix ? (prefix_len < wsize ? prefix_len : wsize) : 0;
    o->prefix = prefix ? prefix + (prefix_len - amount) : NULL;
    o->prefix_amount = amount;
    inflate_body(c, b, trailing_bits, wsize, o, &s->dyn, &s->ff, &s->fn);
}

static void scratch_free(inflate_scratch *s) { free_dhi(&s->dyn); free_dhi(&s->ff); free_dhi(&s->fn); }

void orc_inflate(const uint8_t *in, size_t in_len, uint64_t start_bit,
                 unsigned trailing_bits, uint32_t window_size,
                 const uint8_t *prefix, size_t prefix_len,
                 uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    inflate_scratch s;
    memset(&s, 0, sizeof s);
    memset(res, 0, sizeof *res);
    bitrd b = { in, 8ull * in_len, start_bit, &c };
    outbuf o = { out, out_cap, 0, NULL, 0 };
    if (setjmp(c.jb) == 0) {
        inflate_run(&c, &b, trailing_bits, window_size, prefix, prefix_len, &o, &s);
        res->status = SEMZ_OK;
    } else {
        res->status = c.status;
        res->detail = c.detail;
    }
    res->in_end_bit = b.pos;
    res->out_len = o.n;
    scratch_free(&s);
}

static void put_le32(uint8_t *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }

/* zlib.lisp:21-78 */
void orc_zlib_member(const uint8_t *in, size_t in_len, int have_dict_fn,
                     const uint8_t *dict, size_t dict_len,
                     uint8_t *out, size_t out_cap, orc_zlib_header *hdr, orc_result *res)
{
    orc_ctx c;
    inflate_scratch s;
    memset(&s, 0, sizeof s);
    memset(res, 0, sizeof *res);
    memset(hdr, 0, sizeof *hdr);
    bitrd b = { in, 8ull * in_len, 0, &c };
    outbuf o = { out, out_cap, 0, NULL, 0 };
    if (setjmp(c.jb) == 0) {
        unsigned cmf = (unsigned)lbr_read(&b, 8);       /* bs-read-byte x2 */
        unsigned flg = (unsigned)lbr_read(&b, 8);
        if ((cmf & 15) != 8) orc_die(&c, SEMZ_E_ZLIB_METHOD, cmf & 15);
        if ((cmf >> 4) > 7) orc_die(&c, SEMZ_E_ZLIB_CINFO, cmf >> 4);
        if ((cmf * 256 + flg) % 31) orc_die(&c, SEMZ_E_ZLIB_FCHECK, 0);
        hdr->window_size = 1u << ((cmf >> 4) + 8);
        hdr->level = flg >> 6;
        hdr->has_dict = (flg >> 5) & 1;
        const uint8_t *prefix = NULL;
        size_t prefix_len = 0;
        if (hdr->has_dict) {
            uint32_t id = 0;
            for (int i = 0; i < 4; i++) id = (id << 8) | (uint32_t)lbr_read(&b, 8);   /* bs-read-be 4 */
            hdr->dict_id = id;
            if (!have_dict_fn) orc_die(&c, SEMZ_E_ZLIB_NO_DICT_FN, 0);
            if (!dict) orc_die(&c, SEMZ_E_ZLIB_UNKNOWN_DICT, id);
            prefix = dict;
            prefix_len = dict_len;
        }
        inflate_run(&c, &b, 32, hdr->window_size, prefix, prefix_len, &o, &s);
        br_flush_byte(&b);
        uint32_t want = 0;
        for (int i = 0; i < 4; i++) want = (want << 8) | (uint32_t)lbr_read(&b, 8);   /* zlib.lisp:70-73 */
        uint32_t got = orc_adler32(out, o.n);
        put_le32(res->check, got);
        res->check_len = 4;
        if (want != got) orc_die(&c, SEMZ_E_ZLIB_CHECKSUM, want);
        res->status = SEMZ_OK;
    } else {
        res->status = c.status;
        res->detail = c.detail;
    }
    res->in_end_bit = b.pos;
    res->out_len = o.n;
    scratch_free(&s);
}

/* gzip-extra-fields-consistent-p, gzip.lisp:10-21 */
static int extra_consistent(const uint8_t *p, size_t n)
{
    size_t i = 0;
    while (i < n) {
        if (i + 4 > n) return 0;
        size_t l = p[i + 2] | (size_t)p[i + 3] << 8;
        if (i + 4 + l > n) return 0;
        i += 4 + l;
    }
    return 1;
}

/* gzip.lisp:31-115 */
void orc_gzip_member(const uint8_t *in, size_t in_len,
                     uint8_t *out, size_t out_cap, orc_gzip_header *hdr, orc_result *res)
{
    orc_ctx c;
    inflate_scratch s;
    memset(&s, 0, sizeof s);
    memset(res, 0, sizeof *res);
    memset(hdr, 0, sizeof *hdr);
    hdr->extra_off = hdr->name_off = hdr->comment_off = -1;
    hdr->extra_len = hdr->name_len = hdr->comment_len = -1;
    bitrd b = { in, 8ull * in_len, 0, &c };
    outbuf o = { out, out_cap, 0, NULL, 0 };
    if (setjmp(c.jb) == 0) {
        size_t p = 0;
        if (in_len < 10) orc_die(&c, SEMZ_EOF, 0);
        unsigned flg = in[3];
        if (in[0] != 0x1F || in[1] != 0x8B) orc_die(&c, SEMZ_E_GZIP_MAGIC, 0);
        if (in[2] != 8) orc_die(&c, SEMZ_E_GZIP_METHOD, in[2]);
        if (flg >> 5) orc_die(&c, SEMZ_E_GZIP_RESERVED_FLAGS, 0);
        hdr->textp = flg & 1;
        hdr->mtime = in[4] | (uint32_t)in[5] << 8 | (uint32_t)in[6] << 16 | (uint32_t)in[7] << 24;
        hdr->extra_flags = in[8];
        hdr->os = in[9];
        p = 10;
        if (flg & 4) {
            if (in_len - p < 2) orc_die(&c, SEMZ_EOF, 0);
            size_t xl = in[p] | (size_t)in[p + 1] << 8;
            p += 2;
            if (in_len - p < xl) orc_die(&c, SEMZ_EOF, 0);
            if (!extra_consistent(in + p, xl)) orc_die(&c, SEMZ_E_GZIP_EXTRA_FIELDS, 0);
            hdr->extra_off = (int64_t)p;
            hdr->extra_len = (int64_t)xl;
            p += xl;
        }
        for (int which = 0; which < 2; which++) {       /* FNAME then FCOMMENT, gzip.lisp:71-72 */
            if (!(flg & (which ? 16 : 8))) continue;
            size_t st = p;
            for (;;) {
                if (p >= in_len) orc_die(&c, SEMZ_EOF, 0);
                if (in[p++] == 0) break;
            }
            if (which) { hdr->comment_off = (int64_t)st; hdr->comment_len = (int64_t)(p - 1 - st); }
            else { hdr->name_off = (int64_t)st; hdr->name_len = (int64_t)(p - 1 - st); }
        }
        if (flg & 2) {
            if (in_len - p < 2) orc_die(&c, SEMZ_EOF, 0);
            unsigned want = in[p] | (unsigned)in[p + 1] << 8;
            unsigned got = orc_crc32(in, p) & 0xFFFF;   /* gzip.lisp:73-78 */
            p += 2;
            if (want != got) orc_die(&c, SEMZ_E_GZIP_HEADER_CRC, want);
        }
        hdr->header_len = p;
        b.pos = 8ull * p;
        inflate_run(&c, &b, 64, 32768, NULL, 0, &o, &s);
        br_flush_byte(&b);
        uint32_t want_crc = (uint32_t)lbr_read(&b, 32);
        uint32_t got_crc = orc_crc32(out, o.n);
        put_le32(res->check, got_crc);
        res->check_len = 4;
        if (want_crc != got_crc) orc_die(&c, SEMZ_E_GZIP_DATA_CRC, want_crc);
        uint32_t want_sz = (uint32_t)lbr_read(&b, 32);
        if (want_sz != (uint32_t)o.n) orc_die(&c, SEMZ_E_GZIP_ISIZE, want_sz);
        res->status = SEMZ_OK;
    } else {
        res->status = c.status;
        res->detail = c.detail;
    }
    res->in_end_bit = b.pos;
    res->out_len = o.n;
    scratch_free(&s);
}
