/*
 * bzip2.c -- oracle restatement of bzip2.lisp.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Block pipeline (bzip2.lisp:456-526): block header -> Huffman+RLE2
 * (:222-287) -> inverse MTF (:175-191) -> inverse BWT (:137-163) ->
 * de-randomisation (:112-121) -> RLE1 (:59-102) + block CRC (:507-524).
 *
 * Deliberate divergences from the reference (documented in DESIGN.md):
 *  - origPtr is validated (origPtr < n, or n == 0 with origPtr < block size);
 *    the reference reads a stale array slot instead (bzip2.lisp:158).
 *  - a block whose symbol maps select no symbol at all is SEMZ_E_BZ_NO_SYMBOLS;
 *    the reference only checks the level-1 map (:358-359) and then fails a
 *    Lisp type check in decode-bzip2-huffman+rle2 (:230).
 */
#include "bitio.h"

/* bzip2's randtable.c as quoted in tables.lisp:85-149 */
static const uint16_t RAND_TAB[512] = {
 619,720,127,481,931,816,813,233,566,247,985,724,205,454,863,491,741,242,949,214,
 733,859,335,708,621,574,73,654,730,472,419,436,278,496,867,210,399,680,480,51,
 878,465,811,169,869,675,611,697,867,561,862,687,507,283,482,129,807,591,733,623,
 150,238,59,379,684,877,625,169,643,105,170,607,520,932,727,476,693,425,174,647,
 73,122,335,530,442,853,695,249,445,515,909,545,703,919,874,474,882,500,594,612,
 641,801,220,162,819,984,589,513,495,799,161,604,958,533,221,400,386,867,600,782,
 382,596,414,171,516,375,682,485,911,276,98,553,163,354,666,933,424,341,533,870,
 227,730,475,186,263,647,537,686,600,224,469,68,770,919,190,373,294,822,808,206,
 184,943,795,384,383,461,404,758,839,887,715,67,618,276,204,918,873,777,604,560,
 951,160,578,722,79,804,96,409,713,940,652,934,970,447,318,353,859,672,112,785,
 645,863,803,350,139,93,354,99,820,908,609,772,154,274,580,184,79,626,630,742,
 653,282,762,623,680,81,927,626,789,125,411,521,938,300,821,78,343,175,128,250,
 170,774,972,275,999,639,495,78,352,126,857,956,358,619,580,124,737,594,701,612,
 669,112,134,694,363,992,809,743,168,974,944,375,748,52,600,747,642,182,862,81,
 344,805,988,739,511,655,814,334,249,515,897,955,664,981,649,113,974,459,893,228,
 433,837,553,268,926,240,102,654,459,51,686,754,806,760,493,403,415,394,687,700,
 946,670,656,610,738,392,760,799,887,653,978,321,576,617,626,502,894,679,243,440,
 680,879,194,572,640,724,926,56,204,700,707,151,457,449,797,195,791,558,945,679,
 297,59,87,824,713,663,412,693,342,606,134,108,571,364,631,212,174,643,304,329,
 343,97,430,751,497,314,983,374,822,928,140,206,73,263,980,736,876,478,430,305,
 170,514,364,692,829,82,855,953,676,246,369,970,294,750,807,827,150,790,288,923,
 804,378,215,828,592,281,565,555,710,82,896,831,547,261,524,462,293,465,502,56,
 661,821,976,991,658,869,905,758,745,193,768,550,608,933,378,286,215,979,792,961,
 61,688,793,644,986,403,106,366,905,644,372,567,466,434,645,210,389,550,919,135,
 780,773,635,389,707,100,626,958,165,504,920,176,193,713,857,265,203,50,668,108,
 645,990,626,197,510,357,358,850,858,364,936,638 };

#define BZ_TRAILING_BITS 80     /* +bzip2-trailing-bits+, bzip2.lisp:295-296 */
#define BZ_AB_BOUND 19          /* +bzip2-ab-bound+ = floor(log2(900001)), bzip2.lisp:217-219 */
#define BZ_BLOCK_MAGIC 0x314159265359ull
#define BZ_FOOTER_MAGIC 0x177245385090ull

typedef struct bz_scratch {
    orc_ht sel_tree, trees_tree, trees[6];
    uint8_t *pre, *post;
    uint32_t *perm;
    uint8_t *selectors;
} bz_scratch;

static void bz_free(bz_scratch *s)
{
    free(s->sel_tree.table); free(s->trees_tree.table);
    for (int i = 0; i < 6; i++) free(s->trees[i].table);
    free(s->pre); free(s->post); free(s->perm); free(s->selectors);
    memset(s, 0, sizeof *s);
}

/* decode-bzip2-mtf-in-place, bzip2.lisp:175-191 */
static void mtf_inplace(uint8_t *data, size_t n, const uint8_t *symbols, size_t nsym)
{
    uint8_t stack[256] = {0};
    memcpy(stack, symbols, nsym);
    for (size_t i = 0; i < n; i++) {
        unsigned idx = data[i];
        uint8_t x = stack[idx];
        data[i] = x;
        memmove(stack + 1, stack, idx);
        stack[0] = x;
    }
}

static uint64_t mbr_read48(bitrd *b)                   /* mbr-read-bits 48: 20+20+8, bits.lisp:75-92 */
{
    uint64_t hi = mbr_read(b, 24);
    uint64_t lo = mbr_read(b, 24);
    return (hi << 24) | lo;
}

/* Body of a block after its 48-bit magic: bzip2.lisp:351-395, 463-486, 507-526.
 * Returns the block CRC. */
static uint32_t bz_block_body(orc_ctx *c, bitrd *b, bz_scratch *s, unsigned level,
                              uint8_t *out, size_t cap, size_t *out_len)
{
    const size_t block_size = 100000u * level;
    if (!s->pre) {
        s->pre = (uint8_t *)malloc(900000);
        s->post = (uint8_t *)malloc(900000);
        s->perm = (uint32_t *)calloc(900000, sizeof(uint32_t));
        s->selectors = (uint8_t *)malloc(32768);
        static const uint8_t sel_l[7] = {1, 2, 3, 4, 5, 6, 6};   /* bzip2.lisp:298-300 */
        static const uint8_t tt_l[3] = {2, 2, 1};                /* bzip2.lisp:313-315 */
        ht_build(c, &s->sel_tree, sel_l, 7, BZ_TRAILING_BITS, 0);
        ht_build(c, &s->trees_tree, tt_l, 3, BZ_TRAILING_BITS, 0);
    }
    /* parse-bzip2-block-header, bzip2.lisp:351-395 */
    uint32_t want_crc = (uint32_t)mbr_read(b, 32);
    int randomized = (int)mbr_read(b, 1);
    uint32_t orig = (uint32_t)mbr_read(b, 24);
    unsigned l1 = (unsigned)mbr_read(b, 16);           /* MSB = group 0 */
    if (l1 == 0) orc_die(c, SEMZ_E_BZ_NO_SYMBOLS, 0);
    uint8_t symbols[256];
    unsigned nsym = 0;
    for (int i = 0; i < 16; i++) {
        if (!((l1 >> (15 - i)) & 1)) continue;
        unsigned l2 = (unsigned)mbr_read(b, 16);
        for (int j = 0; j < 16; j++)
            if ((l2 >> (15 - j)) & 1) symbols[nsym++] = (uint8_t)(16 * i + j);
    }
    unsigned ntrees = (unsigned)mbr_read(b, 3);
    unsigned nsel = (unsigned)mbr_read(b, 15);
    if (ntrees < 2 || ntrees > 6) orc_die(c, SEMZ_E_BZ_TREE_COUNT, ntrees);
    if (nsel == 0) orc_die(c, SEMZ_E_BZ_NO_SELECTORS, 0);
    /* read-bzip2-selectors, bzip2.lisp:302-311 */
    for (unsigned i = 0; i < nsel; i++) {
        unsigned v = ht_read(b, &s->sel_tree, 0);
        if (v >= ntrees) orc_die(c, SEMZ_E_BZ_SELECTOR, v);
        s->selectors[i] = (uint8_t)v;
    }
    {
        static const uint8_t six[6] = {0, 1, 2, 3, 4, 5};
        mtf_inplace(s->selectors, nsel, six, 6);
    }
    /* read-bzip2-trees, bzip2.lisp:317-343 */
    for (unsigned t = 0; t < ntrees; t++) {
        uint8_t lens[258];
        int cur = (int)mbr_read(b, 5);
        for (unsigned i = 0; i < nsym + 2; i++) {
            for (;;) {
                if (cur < 1 || cur > 20) orc_die(c, SEMZ_E_BZ_CODELEN, 0);
                unsigned code = ht_read(b, &s->trees_tree, 0);
                if (code == 0) cur++;
                else if (code == 1) cur--;
                else break;
            }
            lens[i] = (uint8_t)cur;
        }
        ht_build(c, &s->trees[t], lens, (int)nsym + 2, BZ_TRAILING_BITS, 0);
    }
    if (nsym == 0) orc_die(c, SEMZ_E_BZ_NO_SYMBOLS, 1);   /* divergence: see file header */

    /* decode-bzip2-huffman+rle2, bzip2.lisp:222-287 */
    size_t n = 0;
    {
        unsigned ab_len = 0, eob = nsym + 1;
        uint32_t ab_val = 0;
        int done = 0;
        for (unsigned si = 0; !done; si++) {
            if (si >= nsel) orc_die(c, SEMZ_E_BZ_SELECTORS_EXHAUSTED, 0);
            const orc_ht *ht = &s->trees[s->selectors[si]];
            for (int k = 0; k < 50 && !done; k++) {
                unsigned code = ht_read(b, ht, 0);
                if (code < 2) {
                    if (ab_len >= BZ_AB_BOUND) orc_die(c, SEMZ_E_BZ_AB_RUN_LENGTH, 0);
                    ab_val |= code << ab_len;
                    ab_len++;
                    continue;
                }
                if (ab_len) {
                    uint32_t run = (ab_val | (1u << ab_len)) - 1;
                    if (n + run > block_size) orc_die(c, SEMZ_E_BZ_AB_RUN_SIZE, 0);
                    memset(s->pre + n, 0, run);
                    n += run;
                    ab_len = 0;
                    ab_val = 0;
                }
                if (code == eob) { done = 1; break; }
                if (n >= block_size) orc_die(c, SEMZ_E_BZ_TOO_MANY_CODES, 0);
                s->pre[n++] = (uint8_t)(code - 1);
            }
        }
    }
    mtf_inplace(s->pre, n, symbols, nsym);
    /* decode-bzip2-bwt, bzip2.lisp:137-163 */
    if (n == 0 ? orig >= block_size : orig >= n) orc_die(c, SEMZ_E_BZ_ORIGPTR, orig);
    {
        uint32_t cumm[256] = {0}, acc = 0;
        for (size_t i = 0; i < n; i++) cumm[s->pre[i]]++;
        for (int v = 0; v < 256; v++) { uint32_t t = cumm[v]; cumm[v] = acc; acc += t; }
        for (size_t i = 0; i < n; i++) s->perm[cumm[s->pre[i]]++] = (uint32_t)i;
        if (n) {
            uint32_t i = s->perm[orig];
            for (size_t j = 0; j < n; j++) { s->post[j] = s->pre[i]; i = s->perm[i]; }
        }
    }
    /* derandomize-bzip2-block, bzip2.lisp:112-121 */
    if (randomized) {
        size_t di = 0;
        for (size_t ri = 0;; ri++) {
            unsigned skip = RAND_TAB[ri % 512];
            di = ri == 0 ? (size_t)skip - 2 : di + skip;
            if (di >= n) break;
            s->post[di] ^= 1;
        }
    }
    /* decode-bzip2-rle1, bzip2.lisp:59-102 */
    size_t o = 0;
    {
        int reps = 0, last = -1;
        for (size_t i = 0; i < n; i++) {
            unsigned by = s->post[i];
            if (reps < 3) {
                reps = ((int)by == last) ? reps + 1 : 0;
                if (o >= cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
                out[o++] = (uint8_t)by;
                last = (int)by;
            } else {
                if (o + by > cap) orc_die(c, SEMZ_E_OUT_CAP, 0);
                memset(out + o, last, by);
                o += by;
                last = -1;
                reps = 0;
            }
        }
    }
    *out_len = o;
    uint32_t crc = orc_bzcrc(out, o);
    if (crc != want_crc) {
        c->detail = want_crc;
        orc_die(c, SEMZ_E_BZ_BLOCK_CRC, want_crc);
    }
    return crc;
}

void orc_bzip2_block(const uint8_t *in, size_t in_len, uint64_t start_bit, unsigned level,
                     uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    bz_scratch s;
    memset(&s, 0, sizeof s);
    memset(res, 0, sizeof *res);
    bitrd b = { in, 8ull * in_len, start_bit, &c };
    size_t n = 0;
    if (setjmp(c.jb) == 0) {
        uint64_t magic = mbr_read48(&b);
        if (magic != BZ_BLOCK_MAGIC) orc_die(&c, SEMZ_E_BZ_BLOCK_MAGIC, 0);
        uint32_t crc = bz_block_body(&c, &b, &s, level, out, out_cap, &n);
        res->check[0] = crc; res->check[1] = crc >> 8; res->check[2] = crc >> 16; res->check[3] = crc >> 24;
        res->check_len = 4;
        res->status = SEMZ_OK;
    } else {
        res->status = c.status;
        res->detail = c.detail;
    }
    res->in_end_bit = b.pos;
    res->out_len = n;
    bz_free(&s);
}

/* byte-source->decompression-state :bzip2 + next-decompressed-chunk, bzip2.lisp:425-526 */
void orc_bzip2_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, orc_result *res)
{
    orc_ctx c;
    bz_scratch s;
    memset(&s, 0, sizeof s);
    memset(res, 0, sizeof *res);
    bitrd b = { in, 8ull * in_len, 0, &c };
    size_t total = 0;
    if (setjmp(c.jb) == 0) {
        unsigned m1 = (unsigned)mbr_read(&b, 8), m2 = (unsigned)mbr_read(&b, 8);
        if (m1 != 0x42 || m2 != 0x5A) orc_die(&c, SEMZ_E_BZ_MAGIC, (m1 << 8) | m2);
        unsigned ver = (unsigned)mbr_read(&b, 8), lvl = (unsigned)mbr_read(&b, 8);
        if (ver != 0x68) orc_die(&c, SEMZ_E_BZ_VERSION, ver);
        if (lvl < 0x31 || lvl > 0x39) orc_die(&c, SEMZ_E_BZ_LEVEL, lvl);
        unsigned level = lvl - 0x30;
        uint32_t stream_crc = 0;
        for (;;) {
            uint64_t magic = mbr_read48(&b);
            if (magic == BZ_BLOCK_MAGIC) {
                size_t n = 0;
                uint32_t crc = bz_block_body(&c, &b, &s, level, out + total, out_cap - total, &n);
                total += n;
                stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;    /* bzip2.lisp:44-47 */
            } else if (magic == BZ_FOOTER_MAGIC) {
                uint32_t want = (uint32_t)mbr_read(&b, 32);
                if (want != stream_crc) orc_die(&c, SEMZ_E_BZ_STREAM_CRC, want);
                br_flush_byte(&b);
                break;
            } else {
                orc_die(&c, SEMZ_E_BZ_BLOCK_MAGIC, 0);
            }
        }
        res->detail = level;
        res->status = SEMZ_OK;
    } else {
        res->status = c.status;
        res->detail = c.detail;
    }
    res->in_end_bit = b.pos;
    res->out_len = total;
    bz_free(&s);
}
