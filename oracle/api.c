/*
 * api.c -- oracle mirror of decompress / decompress-all (interface.lisp:46-208).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The `refill` state machine of interface.lisp:91-133 over a fully buffered
 * input: decode a member; in all-members mode, if input remains, ask the
 * format for a reset state (gzip.lisp:92-93, bzip2.lisp:453-454,
 * xz.lisp:118-134); formats without one fail with "Trailing data detected."
 */
#include "oracle.h"
#include <string.h>

void orc_xz_next_member_start(const uint8_t *in, size_t in_len, size_t pos,
                              size_t *next, int *status);   /* xz.c */
void orc_xz_member_ex(const uint8_t *in, size_t in_len, int skip_magic, uint8_t *out, size_t out_cap,
                      orc_result *res);                     /* xz.c */

void orc_decompress(int fmt, const uint8_t *in, size_t in_len, int all_members,
                    uint32_t opt, uint8_t *out, size_t out_cap, orc_result *res)
{
    size_t pos = 0, outn = 0;
    int skip_magic = 0;
    orc_result r;
    memset(res, 0, sizeof *res);
    for (;;) {
        const uint8_t *p = in + pos;
        size_t n = in_len - pos;
        switch (fmt) {
        case 0: orc_inflate(p, n, 0, 0, opt ? opt : 32768, NULL, 0, out + outn, out_cap - outn, &r); break;
        case 1: { orc_zlib_header h; orc_zlib_member(p, n, 0, NULL, 0, out + outn, out_cap - outn, &h, &r); break; }
        case 2: { orc_gzip_header h; orc_gzip_member(p, n, out + outn, out_cap - outn, &h, &r); break; }
        case 3: orc_bzip2_member(p, n, out + outn, out_cap - outn, &r); break;
        case 5: { uint32_t hdr[5]; uint64_t hs; orc_lzma_alone(p, n, (int)opt, out + outn, out_cap - outn, hdr, &hs, &r); break; }
        case 6: orc_lzma2(p, n, 0, 0, opt, out + outn, out_cap - outn, &r); break;
        case 7: {                                          /* :lzma2 = dictionary size byte + :raw-lzma2, lzma2.lisp:222-227 */
            memset(&r, 0, sizeof r);
            if (n < 1) { r.status = SEMZ_EOF; break; }
            unsigned o = p[0];                             /* parse-lzma2-dict-size, lzma2.lisp:40-46 */
            if (o > 40) { r.status = SEMZ_E_LZMA2_DICT_BYTE; r.detail = o; r.in_end_bit = 8; break; }
            uint32_t ds = o == 40 ? 0xFFFFFFFFu : (2u | (o & 1)) << (11 + o / 2);
            orc_lzma2(p + 1, n - 1, 0, 0, ds, out + outn, out_cap - outn, &r);
            r.in_end_bit += 8;
            break;
        }
        case 8: orc_xz_member_ex(p, n, skip_magic, out + outn, out_cap - outn, &r); break;
        default: r.status = SEMZ_E_UNSUPPORTED; r.detail = 0; r.in_end_bit = 0; r.out_len = 0;
        }
        size_t used = (size_t)((r.in_end_bit + 7) >> 3);
        if (r.status != SEMZ_OK) {
            *res = r;
            res->in_end_bit = 8ull * (pos + used);
            res->out_len = outn;
            return;
        }
        pos += used;
        outn += r.out_len;
        if (!all_members || pos == in_len) break;      /* buffer-stream-check-eof */
        if (fmt == 2 || fmt == 3) continue;            /* concatenated members */
        if (fmt == 8) {
            size_t next; int st;
            orc_xz_next_member_start(in, in_len, pos, &next, &st);
            if (st != SEMZ_OK) { res->status = st; res->in_end_bit = 8ull * pos; res->out_len = outn; return; }
            pos = next;
            if (pos == in_len) break;
            skip_magic = 1;
            continue;
        }
        res->status = SEMZ_E_TRAILING_DATA;
        res->in_end_bit = 8ull * pos;
        res->out_len = outn;
        return;
    }
    *res = r;
    res->status = SEMZ_OK;
    res->in_end_bit = 8ull * pos;
    res->out_len = outn;
}
