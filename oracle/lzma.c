/*
 * lzma.c -- oracle restatement of lzma.lisp and lzma2.lisp.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows, function by function:
 *   byte sources with limits          io.lisp:258-323 (counted buffer streams)
 *   lzma-range-decoder                lzma.lisp:49-198
 *   decode-lzma-length / -distance    lzma.lisp:247-341
 *   decode-lzma (symbol loop)         lzma.lisp:566-775, all three eof modes
 *   :raw-lzma / :lzma constructors    lzma.lisp:809-865, chunk method 867-895
 *   LZMA2 control-byte state machine  lzma2.lisp:77-204
 *
 * The reference decodes into a sliding dictionary buffer and hands out
 * chunks; here the whole output buffer is the dictionary, which is
 * indistinguishable as long as the checks of handle-match (lzma.lisp:600-612)
 * are applied to "bytes since the last dictionary reset" (the reference's
 * buffer-i before any flush; after a flush buffer-i >= dict-size and the
 * dict-size check, which comes first, subsumes it).
 */
#include "lzma_src.h"

/* ---- range decoder (lzma.lisp:49-198) ---- */
typedef struct lrd {
    uint32_t code, range;
    osrc *s;
} lrd;

static void lrd_init(lrd *r, osrc *s)                       /* lzma.lisp:55-62 */
{
    r->s = s;
    if (osrc_byte(s) != 0) orc_die(s->ctx, SEMZ_E_LZMA_FIRST_BYTE, 0);
    r->range = 0xFFFFFFFFu;
    r->code = (uint32_t)osrc_be(s, 4);
    if (!(r->code < r->range)) orc_die(s->ctx, SEMZ_E_LZMA_CODE_START, 0);
}

static inline void lrd_normalize(lrd *r)                    /* lzma.lisp:64-73 */
{
    if (r->range < (1u << 24)) {
        r->range <<= 8;
        r->code = (r->code << 8) | osrc_byte(r->s);
        if (!(r->code < r->range)) orc_die(r->s->ctx, SEMZ_E_LZMA_CODE_RANGE, 0);
    }
}

static inline unsigned lrd_bit(lrd *r, uint16_t *p)         /* lzma.lisp:83-116 */
{
    uint32_t zp = *p, bound = zp * (r->range >> 11);
    unsigned bit;
    if (r->code >= bound) {
        r->range -= bound;
        r->code -= bound;
        *p = (uint16_t)(zp - (zp >> 5));
        bit = 1;
    } else {
        r->range = bound;
        *p = (uint16_t)(zp + ((2048 - zp) >> 5));
        bit = 0;
    }
    lrd_normalize(r);
    return bit;
}

static inline unsigned lrd_be_bits(lrd *r, uint16_t *probs, unsigned count)   /* lzma.lisp:122-135 */
{
    unsigned bound = 1u << count, m = 1;
    while (m < bound) m = (m << 1) | lrd_bit(r, probs + m);
    return m - bound;
}

static inline unsigned rev_small(unsigned x, unsigned n)
{
    unsigned v = 0;
    for (unsigned i = 0; i < n; i++) v |= ((x >> i) & 1) << (n - 1 - i);
    return v;
}

static inline unsigned lrd_le_bits(lrd *r, uint16_t *probs, unsigned count)   /* lzma.lisp:138-145 */
{
    return rev_small(lrd_be_bits(r, probs, count), count);
}

static inline uint32_t lrd_fixed_bits(lrd *r, unsigned count)                 /* lzma.lisp:174-192 */
{
    uint32_t res = 0;
    for (unsigned i = 0; i < count; i++) {
        uint32_t bound = r->range >> 1;
        r->range = bound;
        if (r->code >= bound) {
            r->code -= bound;
            if (r->code == bound) orc_die(r->s->ctx, SEMZ_E_LZMA_CODE_RANGE, 0);
            res = (res << 1) | 1;
        } else {
            res <<= 1;
        }
        lrd_normalize(r);
    }
    return res;
}

/* ---- model (lzma.lisp:220-460) ---- */
typedef struct lendec {
    uint16_t choices[2];
    uint16_t low[16][8], mid[16][8];
    uint16_t high[256];
} lendec;

typedef struct lzvars {
    unsigned history;                /* symbol-history, 0..11 */
    uint16_t is_literal[12 << 4], is_simple_match[12], uses_rep0[12], is_short_rep[12 << 4],
             is_rep1[12], is_rep2[12];
    lendec simple_len, rep_len;
    uint16_t slot[4][64], mid[115], alignment[16];
    uint32_t rep0, rep1, rep2, rep3; /* distance codes */
    unsigned total_i_mod_16;
    size_t nliteral;                 /* 0x300 << (lc + lp): lc + lp <= 4 in LZMA2, <= 12 in LZMA */
    uint16_t literal[];
} lzvars;

static void probs_fill(uint16_t *p, size_t n) { for (size_t i = 0; i < n; i++) p[i] = 1024; }

static lzvars *lzvars_alloc(unsigned lclp)
{
    lzvars *v = (lzvars *)malloc(sizeof(lzvars) + sizeof(uint16_t) * ((size_t)0x300 << lclp));
    v->nliteral = (size_t)0x300 << lclp;
    return v;
}

static void lzvars_reset(lzvars *v)                          /* make-lzma-state-vars */
{
    size_t nl = v->nliteral;
    memset(v, 0, sizeof *v);
    v->nliteral = nl;
    probs_fill(v->is_literal, 12 << 4); probs_fill(v->is_simple_match, 12); probs_fill(v->uses_rep0, 12);
    probs_fill(v->is_short_rep, 12 << 4); probs_fill(v->is_rep1, 12); probs_fill(v->is_rep2, 12);
    probs_fill(&v->simple_len.choices[0], sizeof(lendec) / 2);
    probs_fill(&v->rep_len.choices[0], sizeof(lendec) / 2);
    probs_fill(&v->slot[0][0], 4 * 64); probs_fill(v->mid, 115); probs_fill(v->alignment, 16);
    probs_fill(v->literal, v->nliteral);
}

static unsigned decode_length(lendec *d, lrd *r, unsigned pos_state)          /* lzma.lisp:247-257 */
{
    if (lrd_bit(r, &d->choices[0]) == 0) return lrd_be_bits(r, d->low[pos_state], 3);
    if (lrd_bit(r, &d->choices[1]) == 0) return 8 + lrd_be_bits(r, d->mid[pos_state], 3);
    return 16 + lrd_be_bits(r, d->high, 8);
}

static uint32_t decode_distance(lrd *r, lzvars *v, unsigned raw_len)          /* lzma.lisp:323-341 */
{
    unsigned slot = lrd_be_bits(r, v->slot[raw_len < 3 ? raw_len : 3], 6);
    if (slot < 4) return slot;
    unsigned bit_count = (slot >> 1) - 1;
    uint32_t min_dist = (2u | (slot & 1)) << bit_count;
    if (slot < 14)
        /* index origin min_dist - slot; lrd_be_bits addresses probs[origin + m], m >= 1 */
        return min_dist + lrd_le_bits(r, v->mid + (min_dist - slot), bit_count);
    uint32_t hi = lrd_fixed_bits(r, bit_count - 4) << 4;
    return min_dist + hi + lrd_le_bits(r, v->alignment, 4);
}

static unsigned decode_literal_match(lrd *r, uint16_t *probs, unsigned match_byte)   /* lzma.lisp:497-518 */
{
    unsigned symbol = 1;
    for (int bit_i = 7; bit_i >= 0; bit_i--) {
        unsigned match_bit = (match_byte >> bit_i) & 1;
        unsigned bit = lrd_bit(r, probs + 0x100 + (match_bit << 8) + symbol);
        symbol = (symbol << 1) | bit;
        if (match_bit != bit) break;
    }
    while (symbol < 0x100) symbol = (symbol << 1) | lrd_bit(r, probs + symbol);
    return symbol & 0xFF;
}

typedef struct lzprops { unsigned lc, lp, pb; } lzprops;

enum { EOF_MAYBE = 0, EOF_ALWAYS = 1, EOF_NEVER = 2 };

/*
 * decode-lzma (lzma.lisp:566-775).  out[dict_base .. *pos) is the dictionary.
 * `end` is the variable P of lzma.lisp:532-551 for a declared size, UINT64_MAX
 * when the size is unknown (then only a marker ends the data; running out of
 * out_cap is SEMZ_E_OUT_CAP, an artefact of the flat buffer).  Returns eofp.
 */
static int decode_lzma(orc_ctx *c, lrd *r, const lzprops *pr, lzvars *v, uint32_t dict_size,
                       uint8_t *out, uint64_t out_cap, uint64_t dict_base, uint64_t *pos, uint64_t end, int eof_mode)
{
    static const uint8_t lit_next[12] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 4, 5};
    uint32_t rep0 = v->rep0, rep1 = v->rep1, rep2 = v->rep2, rep3 = v->rep3;
    unsigned history = v->history;
    uint64_t total_i = v->total_i_mod_16, p = *pos;
    int eofp = 0;
    const unsigned lp_mask = (1u << pr->lp) - 1, pb_mask = (1u << pr->pb) - 1, lc = pr->lc;

    for (;;) {
        unsigned pos_state = (unsigned)total_i & pb_mask;
        if (p == end) {
            /* handle-eof-insanity (lzma.lisp:640-659): :always probes at once, the other two ask the range decoder first */
            if (eof_mode == EOF_ALWAYS || r->code != 0) {
                int marker = 0;                             /* probe-eof, lzma.lisp:623-638 */
                if (lrd_bit(r, &v->is_literal[(history << 4) | pos_state]) == 1 &&
                    lrd_bit(r, &v->is_simple_match[history]) == 0) {
                    unsigned len = decode_length(&v->simple_len, r, pos_state);
                    history = history < 7 ? 7 : 10;
                    rep3 = rep2; rep2 = rep1; rep1 = rep0;
                    rep0 = decode_distance(r, v, len);
                    marker = rep0 == 0xFFFFFFFFu;
                }
                if (eof_mode == EOF_NEVER) orc_die(c, marker ? SEMZ_E_LZMA_ILLEGAL_EOF : SEMZ_E_LZMA_CONTINUES, 0);
                if (!marker) orc_die(c, eof_mode == EOF_ALWAYS ? SEMZ_E_LZMA_NO_EOF_MARKER : SEMZ_E_LZMA_NO_END, 0);
            }
            eofp = 1;
            break;
        }
        unsigned length;
        if (lrd_bit(r, &v->is_literal[(history << 4) | pos_state]) == 0) {
            /* decode-literal, lzma.lisp:661-688 */
            unsigned prev = p == dict_base ? 0 : out[p - 1];
            uint16_t *probs = v->literal + (size_t)0x300 * ((((unsigned)total_i & lp_mask) << lc) | (prev >> (8 - lc)));
            unsigned byte;
            if (history < 7) byte = lrd_be_bits(r, probs, 8);
            else {
                if (p - dict_base < (uint64_t)rep0 + 1) orc_die(c, SEMZ_E_UNSUPPORTED, 1);   /* assert, lzma.lisp:687 */
                byte = decode_literal_match(r, probs, out[p - 1 - rep0]);
            }
            if (p >= out_cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
            out[p++] = (uint8_t)byte;
            total_i++;
            history = lit_next[history];
            continue;
        }
        if (lrd_bit(r, &v->is_simple_match[history]) == 0) {
            unsigned len = decode_length(&v->simple_len, r, pos_state);
            history = history < 7 ? 7 : 10;
            rep3 = rep2; rep2 = rep1; rep1 = rep0;
            rep0 = decode_distance(r, v, len);
            if (rep0 == 0xFFFFFFFFu) {                                         /* lzma.lisp:727-732 */
                if (eof_mode == EOF_NEVER) orc_die(c, SEMZ_E_LZMA_ILLEGAL_EOF, 0);
                eofp = 1;
                break;
            }
            length = len + 2;
        } else if (lrd_bit(r, &v->uses_rep0[history]) == 0) {
            if (lrd_bit(r, &v->is_short_rep[(history << 4) | pos_state]) == 0) {
                history = history < 7 ? 9 : 11;
                length = 1;
            } else {
                history = history < 7 ? 8 : 11;
                length = 2 + decode_length(&v->rep_len, r, pos_state);
            }
        } else {
            uint32_t t;
            if (lrd_bit(r, &v->is_rep1[history]) == 0) { t = rep1; rep1 = rep0; rep0 = t; }
            else if (lrd_bit(r, &v->is_rep2[history]) == 0) { t = rep2; rep2 = rep1; rep1 = rep0; rep0 = t; }
            else { t = rep3; rep3 = rep2; rep2 = rep1; rep1 = rep0; rep0 = t; }
            history = history < 7 ? 8 : 11;
            length = 2 + decode_length(&v->rep_len, r, pos_state);
        }
        /* handle-match, lzma.lisp:600-612; distance = rep0 + 1 (may be 2^32) */
        uint64_t distance = (uint64_t)rep0 + 1;
        if (!(distance <= dict_size)) orc_die(c, SEMZ_E_LZMA_DICT_SIZE, 0);
        if (!(distance <= p - dict_base)) orc_die(c, SEMZ_E_LZMA_NO_DATA, 0);
        if (!(length <= end - p)) orc_die(c, SEMZ_E_LZMA_MATCH_SIZE, 0);
        if (p + length > out_cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
        for (unsigned i = 0; i < length; i++) out[p + i] = out[p + i - distance];   /* copy-match, common.lisp:324-333 */
        p += length;
        total_i += length;
    }
    v->rep0 = rep0; v->rep1 = rep1; v->rep2 = rep2; v->rep3 = rep3;
    v->history = history;
    v->total_i_mod_16 = (unsigned)(total_i & 15);
    *pos = p;
    if (eofp && r->code != 0) orc_die(c, SEMZ_E_LZMA_FINISH_STATE, 0);     /* lzma.lisp:772-774 */
    return eofp;
}

/* parse-lzma-props + parse-lzma2-props, lzma.lisp:837-843, lzma2.lisp:31-38 */
static void parse_lzma2_props(orc_ctx *c, unsigned octet, lzprops *pr)
{
    pr->lc = octet % 9;
    pr->lp = (octet / 9) % 5;
    pr->pb = octet / 45;
    if (pr->pb > 4) orc_die(c, SEMZ_E_LZMA_PB, pr->pb);
    if (pr->lc + pr->lp > 4) orc_die(c, SEMZ_E_LZMA2_LCLP, pr->lc + pr->lp);
}

/*
 * LZMA2 chunk sequence (lzma2.lisp:77-204) from `s` into out[*pos ..), until
 * the end control byte.  The dictionary starts empty at *pos (buffer-i 0).
 */
void orc_lzma2_run(orc_ctx *c, osrc *s, uint32_t dict_size, uint8_t *out, size_t out_cap, uint64_t *pos)
{
    enum { NEED_DICT, NEED_PROPS, NEED_NONE } need = NEED_DICT;
    lzvars *v = lzvars_alloc(4);
    lzprops props = {0, 0, 0};
    int have_state = 0;
    uint64_t dict_base = *pos;
    orc_ctx inner;
    inner.status = SEMZ_OK;
    inner.detail = 0;
    /* errors unwind through here so that `v` is released */
    osrc_rebind(s, &inner);
    if (setjmp(inner.jb)) {
        free(v);
        osrc_rebind(s, c);
        orc_die(c, inner.status, inner.detail);
    }
    for (;;) {
        unsigned control = osrc_byte(s);
        if (control >= 0x80) {
            uint64_t usize = ((uint64_t)(control & 31) << 16 | osrc_be(s, 2)) + 1;
            uint64_t csize = osrc_be(s, 2) + 1;
            if (need == NEED_DICT && control < 0xE0) orc_die(&inner, SEMZ_E_LZMA2_NEED_DICT_RESET, 0);
            if (need == NEED_PROPS && control < 0xC0) orc_die(&inner, SEMZ_E_LZMA2_NEED_PROPS, 0);
            need = NEED_NONE;
            if (control >= 0xC0) parse_lzma2_props(&inner, osrc_byte(s), &props);
            else if (!have_state) orc_die(&inner, SEMZ_E_LZMA2_NO_STATE, 0);   /* (lzs-properties nil), lzma2.lisp:107 */
            if (control >= 0xE0) dict_base = *pos;
            if (*pos + usize > out_cap) orc_die(&inner, SEMZ_E_OUT_CAP, 0);
            osrc_push_limit(s, csize, SEMZ_E_LZMA2_CHUNK_OVERRUN);
            lrd r;
            lrd_init(&r, s);
            if (control >= 0xA0 || !have_state) lzvars_reset(v);             /* lzma2.lisp:144-145 */
            have_state = 1;
            decode_lzma(&inner, &r, &props, v, dict_size, out, out_cap, dict_base, pos, *pos + usize, EOF_NEVER);
            if (osrc_pop_limit(s) != 0) orc_die(&inner, SEMZ_E_LZMA2_CHUNK_SHORT, 0);   /* lzma2.lisp:200-201 */
        } else if (control > 2) {
            orc_die(&inner, SEMZ_E_LZMA2_CONTROL, control);
        } else if (control >= 1) {
            uint64_t usize = osrc_be(s, 2) + 1;
            if (control == 2) {
                if (need == NEED_DICT) orc_die(&inner, SEMZ_E_LZMA2_NEED_DICT_RESET, 0);
                need = NEED_NONE;
            } else {
                need = NEED_PROPS;
                dict_base = *pos;
            }
            if (*pos + usize > out_cap) orc_die(&inner, SEMZ_E_OUT_CAP, 0);
            osrc_read(s, out + *pos, usize);
            *pos += usize;
            if (have_state) v->total_i_mod_16 = (unsigned)((v->total_i_mod_16 + usize) & 15);   /* lzma2.lisp:184-188 */
        } else {
            break;
        }
    }
    free(v);
    osrc_rebind(s, c);
}

void orc_lzma2(const uint8_t *in, size_t in_len, int have_limit, uint64_t limit, uint32_t dict_size,
               uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    osrc s;
    uint64_t pos = 0;
    memset(res, 0, sizeof *res);
    osrc_init(&s, &c, in, in_len, 0);
    if (setjmp(c.jb)) {
        res->status = c.status;
        res->detail = c.detail;
        res->in_end_bit = 8 * s.pos;
        res->out_len = pos;
        return;
    }
    if (have_limit) osrc_push_limit(&s, limit, SEMZ_E_XZ_BLOCK_OVERRUN);
    orc_lzma2_run(&c, &s, dict_size, out, out_cap, &pos);
    res->status = SEMZ_OK;
    res->in_end_bit = 8 * s.pos;
    res->out_len = pos;
}

/*
 * :raw-lzma (lzma.lisp:809-835, 867-895): range decoder start, one decode-lzma
 * run over the whole output (chunking has no effect on the checks, see above).
 * size = declared decompressed size when have_size.
 */
void orc_lzma(const uint8_t *in, size_t in_len, unsigned lc, unsigned lp, unsigned pb, uint32_t dict_size,
              int have_size, uint64_t size, int eof_mode, uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    osrc s;
    uint64_t pos = 0;
    lzvars *v = lzvars_alloc(lc + lp);
    memset(res, 0, sizeof *res);
    osrc_init(&s, &c, in, in_len, 0);
    if (setjmp(c.jb)) {
        free(v);
        res->status = c.status;
        res->detail = c.detail;
        res->in_end_bit = 8 * s.pos;
        res->out_len = pos;
        return;
    }
    lrd r;
    lrd_init(&r, &s);
    lzprops pr = {lc, lp, pb};
    lzvars_reset(v);
    uint64_t end = have_size ? size : UINT64_MAX;
    int eofp = decode_lzma(&c, &r, &pr, v, dict_size, out, out_cap, 0, &pos, end, eof_mode);
    if (eofp && have_size && pos != size) orc_die(&c, SEMZ_E_LZMA_EARLY_END, 0);    /* lzma.lisp:891-894 */
    free(v);
    res->status = SEMZ_OK;
    res->in_end_bit = 8 * s.pos;
    res->out_len = pos;
}

/*
 * :lzma, the .lzma "alone" container (lzma.lisp:837-865).  As written the
 * constructor hands the declared size on under the keyword
 * :expected-remaining-data, which the :raw-lzma method does not have (its
 * keyword is :decompressed-size, lzma.lisp:811, 857), so the size only reaches
 * the header plist and the data always has to end with a marker; honour_size
 * = 1 restates what the code evidently intends (size + eof-mode :maybe).
 * hdr[0..4] = lc, lp, pb, window size, 1 if a size is declared; *hsize = it.
 */
void orc_lzma_alone(const uint8_t *in, size_t in_len, int honour_size, uint8_t *out, size_t out_cap,
                    uint32_t hdr[5], uint64_t *hsize, orc_result *res)
{
    memset(res, 0, sizeof *res);
    memset(hdr, 0, 5 * sizeof hdr[0]);
    *hsize = 0;
    if (in_len < 1) { res->status = SEMZ_EOF; return; }
    unsigned octet = in[0], lc = octet % 9, lp = (octet / 9) % 5, pb = octet / 45;   /* parse-lzma-props */
    if (pb > 4) { res->status = SEMZ_E_LZMA_PB; res->detail = pb; res->in_end_bit = 8; return; }
    if (in_len < 13) { res->status = SEMZ_EOF; res->in_end_bit = 8 * in_len; return; }
    uint32_t dict = 0;
    uint64_t size = 0;
    for (int i = 0; i < 4; i++) dict |= (uint32_t)in[1 + i] << (8 * i);
    for (int i = 0; i < 8; i++) size |= (uint64_t)in[5 + i] << (8 * i);
    int have = size != UINT64_MAX;
    hdr[0] = lc; hdr[1] = lp; hdr[2] = pb; hdr[3] = dict; hdr[4] = (uint32_t)have;
    *hsize = size;
    orc_lzma(in + 13, in_len - 13, lc, lp, pb, dict, have && honour_size, size, EOF_MAYBE, out, out_cap, res);
    res->in_end_bit += 8 * 13;
}
