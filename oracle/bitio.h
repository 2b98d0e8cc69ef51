/*
 * bitio.h -- oracle bit readers and canonical Huffman tables.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates bits.lisp:18-121 (lsb-bit-reader / msb-bit-reader) and
 * huffman.lisp:71-225 (huffman-tree, ht-read-le-code / ht-read-be-code,
 * lengths->ht) over a fully buffered input.  The reference refills one byte
 * at a time, so "ensure n bits" fails with `eof` exactly when fewer than n
 * bits remain between the read position and the end of the input; that is the
 * rule implemented by br_ensure().
 */
#ifndef SEMZ_ORACLE_BITIO_H
#define SEMZ_ORACLE_BITIO_H

#include <setjmp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct orc_ctx {
    jmp_buf jb;
    int status;
    uint32_t detail;
} orc_ctx;

static inline void orc_die(orc_ctx *c, int status, uint32_t detail)
{
    c->status = status;
    c->detail = detail;
    longjmp(c->jb, 1);
}

typedef struct bitrd {
    const uint8_t *in;
    uint64_t nbits;   /* 8 * input length */
    uint64_t pos;     /* bit position */
    orc_ctx *ctx;
} bitrd;

static inline void br_ensure(bitrd *b, unsigned n)     /* bits.lisp:35-45 */
{
    if (b->nbits - b->pos < n) orc_die(b->ctx, SEMZ_EOF, 0);
}

static inline unsigned br_bit(const bitrd *b, uint64_t p, int msb)
{
    if (p >= b->nbits) return 0;                       /* zero padding, bits.lisp:49-50 */
    unsigned byte = b->in[p >> 3];
    return msb ? (byte >> (7 - (p & 7))) & 1 : (byte >> (p & 7)) & 1;
}

/* LSB-first: bit i of the result is stream bit pos+i (bits.lisp:52) */
static inline uint64_t lbr_peek(const bitrd *b, unsigned n)
{
    uint64_t byte = b->pos >> 3;
    if ((byte + 8) * 8 <= b->nbits && n <= 56) {       /* fast path: whole window in bounds */
        uint64_t w;
        memcpy(&w, b->in + byte, 8);
        return (w >> (b->pos & 7)) & ((n == 64) ? ~0ull : ((1ull << n) - 1));
    }
    uint64_t v = 0;
    for (unsigned i = 0; i < n; i++) v |= (uint64_t)br_bit(b, b->pos + i, 0) << i;
    return v;
}
static inline uint64_t lbr_read(bitrd *b, unsigned n)  /* bits.lisp:65-92 */
{
    br_ensure(b, n);
    uint64_t v = lbr_peek(b, n);
    b->pos += n;
    return v;
}
/* MSB-first: first stream bit is the most significant (bits.lisp:53) */
static inline uint64_t mbr_peek(const bitrd *b, unsigned n)
{
    uint64_t byte = b->pos >> 3;
    if ((byte + 8) * 8 <= b->nbits && n >= 1 && n <= 56) {
        uint64_t w;
        memcpy(&w, b->in + byte, 8);
        w = __builtin_bswap64(w);
        return (w << (b->pos & 7)) >> (64 - n);
    }
    uint64_t v = 0;
    for (unsigned i = 0; i < n; i++) v = (v << 1) | br_bit(b, b->pos + i, 1);
    return v;
}
static inline uint64_t mbr_read(bitrd *b, unsigned n)
{
    br_ensure(b, n);
    uint64_t v = mbr_peek(b, n);
    b->pos += n;
    return v;
}
static inline void br_flush_byte(bitrd *b)             /* bits.lisp:94-106 */
{
    b->pos = (b->pos + 7) & ~7ull;
}

/* huffman.lisp:71-79; entry = (item << 5) | len (huffman.lisp:53-68) */
typedef struct orc_ht {
    uint16_t *table;
    size_t cap;
    int min_len, max_len, full_read;
} orc_ht;

static inline unsigned rev_bits(unsigned x, unsigned n) /* huffman.lisp:115-126 */
{
    unsigned r = 0;
    for (unsigned i = 0; i < n; i++) r |= ((x >> i) & 1) << (n - 1 - i);
    return r;
}

/* lengths->ht, huffman.lisp:142-225 */
static inline void ht_build(orc_ctx *c, orc_ht *ht, const uint8_t *lengths, int n,
                            long max_overread_bits, int le)
{
    int maxl = 1, minl = 20;
    for (int i = 0; i < n; i++) {
        if (lengths[i] > maxl) maxl = lengths[i];
        int k = lengths[i] ? lengths[i] : 20;
        if (k < minl) minl = k;
    }
    uint64_t unusable = 0, full = 1ull << maxl;
    for (int i = 0; i < n; i++)
        if (lengths[i]) unusable += 1ull << (maxl - lengths[i]);
    if (unusable < full) orc_die(c, SEMZ_E_HUFF_UNDERFULL, 0);
    if (unusable > full) orc_die(c, SEMZ_E_HUFF_OVERFULL, 0);
    if (ht->cap < full) {
        free(ht->table);
        ht->table = (uint16_t *)malloc(full * sizeof(uint16_t));
        ht->cap = full;
    }
    unsigned next = 0;
    for (int len = minl; len <= maxl; len++) {
        next <<= 1;
        for (int item = 0; item < n; item++) {
            if (lengths[item] != len) continue;
            uint16_t e = (uint16_t)((item << 5) | len);
            if (le) {
                for (uint64_t i = rev_bits(next, len); i < full; i += 1ull << len) ht->table[i] = e;
            } else {
                unsigned unused = maxl - len;
                for (uint64_t i = (uint64_t)next << unused; i < ((uint64_t)next + 1) << unused; i++)
                    ht->table[i] = e;
            }
            next++;
        }
    }
    ht->min_len = minl;
    ht->max_len = maxl;
    ht->full_read = max_overread_bits >= (maxl - minl);   /* huffman.lisp:163 */
}

/* ht-read-le-code / ht-read-be-code, huffman.lisp:81-113 */
static inline unsigned ht_read(bitrd *b, const orc_ht *ht, int le)
{
    if (ht->full_read) {
        br_ensure(b, ht->max_len);
        uint64_t idx = le ? lbr_peek(b, ht->max_len) : mbr_peek(b, ht->max_len);
        uint16_t e = ht->table[idx];
        b->pos += e & 31;
        return e >> 5;
    }
    for (int el = ht->min_len; el <= ht->max_len; el++) {
        br_ensure(b, el);
        uint64_t idx = le ? lbr_peek(b, el) : mbr_peek(b, el) << (ht->max_len - el);
        uint16_t e = ht->table[idx];
        if ((e & 31) <= el) {
            b->pos += e & 31;
            return e >> 5;
        }
    }
    abort(); /* "Corrupt Huffman tree." -- unreachable for complete trees */
}

#endif
