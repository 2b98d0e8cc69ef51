/*
 * lzma_src.h -- oracle byte source with nested read limits.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates what a fully buffered input looks like through the reference's
 * byte sources (io.lisp:10-129) and counted byte sources (io.lisp:258-323):
 * a read past the innermost limit that has been reached calls that limit's
 * `on-eof` (a `die` with the limit's message); a read past the physical end
 * of the input before any limit is reached signals `eof`
 * (io.lisp:284-322: "count < limit" -> try-refill of the wrapped source).
 */
#ifndef SEMZ_ORACLE_LZMA_SRC_H
#define SEMZ_ORACLE_LZMA_SRC_H

#include "bitio.h"

typedef struct osrc {
    const uint8_t *in;
    uint64_t pos, end;
    uint64_t lim_end[4];   /* absolute positions, innermost last */
    uint64_t lim_start[4];
    int lim_status[4];
    int nlim;
    orc_ctx *ctx;
} osrc;

static inline void osrc_init(osrc *s, orc_ctx *c, const uint8_t *in, uint64_t end, uint64_t pos)
{
    s->in = in; s->pos = pos; s->end = end; s->nlim = 0; s->ctx = c;
}
static inline void osrc_rebind(osrc *s, orc_ctx *c) { s->ctx = c; }

static inline void osrc_push_limit(osrc *s, uint64_t limit, int status)
{
    s->lim_start[s->nlim] = s->pos;
    s->lim_end[s->nlim] = s->pos + limit;
    s->lim_status[s->nlim] = status;
    s->nlim++;
}
/* returns limit - count (0 = everything was read) */
static inline uint64_t osrc_pop_limit(osrc *s)
{
    s->nlim--;
    return s->lim_end[s->nlim] - s->pos;
}
static inline uint64_t osrc_count(const osrc *s) { return s->pos - s->lim_start[s->nlim - 1]; }

static inline void osrc_fail(osrc *s)
{
    for (int i = s->nlim - 1; i >= 0; i--)
        if (s->pos >= s->lim_end[i]) orc_die(s->ctx, s->lim_status[i], 0);
    orc_die(s->ctx, SEMZ_EOF, 0);
}
static inline unsigned osrc_byte(osrc *s)
{
    if (s->pos >= s->end) osrc_fail(s);
    for (int i = s->nlim - 1; i >= 0; i--)
        if (s->pos >= s->lim_end[i]) osrc_fail(s);
    return s->in[s->pos++];
}
static inline uint64_t osrc_be(osrc *s, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 8) | osrc_byte(s);
    return v;
}
static inline uint64_t osrc_le(osrc *s, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint64_t)osrc_byte(s) << (8 * i);
    return v;
}
/* bs-read-sequence with eof-error-p */
static inline void osrc_read(osrc *s, uint8_t *dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) dst[i] = (uint8_t)osrc_byte(s);
}
static inline uint64_t osrc_left(const osrc *s) { return s->end - s->pos; }

void orc_lzma2_run(orc_ctx *c, osrc *s, uint32_t dict_size, uint8_t *out, size_t out_cap, uint64_t *pos);

#endif
