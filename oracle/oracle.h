/*
 * oracle.h -- CPU restatement of se-mz/decompress's decode path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (decompress_amd/,
 * the C-ABI library) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker the HIP path is compared against.
 *
 * Pinning: checked against every known-answer vector the reference ships
 * (example.lisp:6-78, README.org:306-308) by tests/test_oracle_golden.py; valid
 * streams are additionally cross-checked against CPython zlib/bz2/lzma and
 * hashlib.  Reject paths have no vectors in the reference; they follow the
 * source text cited next to each function ("parity pinned by source reading").
 */
#ifndef SEMZ_ORACLE_H
#define SEMZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/semz_status.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- checksums (crc.lisp, adler32.lisp, sha256.lisp, bzip2.lisp:13-49) ---- */
uint32_t orc_crc32_update(uint32_t state, const uint8_t *p, size_t n);   /* state starts 0xFFFFFFFF, finish = ~ */
uint32_t orc_crc32(const uint8_t *p, size_t n);
uint64_t orc_crc64_update(uint64_t state, const uint8_t *p, size_t n);
uint64_t orc_crc64(const uint8_t *p, size_t n);
uint32_t orc_adler32_update(uint32_t state, const uint8_t *p, size_t n); /* state starts 1 */
uint32_t orc_adler32(const uint8_t *p, size_t n);
uint32_t orc_bzcrc_update(uint32_t state, const uint8_t *p, size_t n);   /* MSB-first form, starts 0xFFFFFFFF, finish = ~ */
uint32_t orc_bzcrc(const uint8_t *p, size_t n);
void     orc_sha256(const uint8_t *p, size_t n, uint8_t digest[32]);

/* ---- result of decoding one unit ---- */
typedef struct orc_result {
    int32_t  status;      /* semz_status */
    uint32_t detail;
    uint64_t in_end_bit;  /* bit offset (from `in`) one past the last consumed bit */
    uint64_t out_len;
    uint8_t  check[32];   /* computed checksum as the LE byte string the reference compares */
    uint32_t check_len;
    uint32_t pad;
} orc_result;

/* ---- deflate family (bits.lisp, huffman.lisp, deflate.lisp, zlib.lisp, gzip.lisp) ---- */

/* Raw deflate member starting at bit `start_bit` of in[0..in_len).  trailing_bits
 * is the reference's `trailing-bits` argument (0 raw, 32 zlib, 64 gzip;
 * deflate.lisp:365).  prefix = preset history (deflate.lisp:419-425). */
void orc_inflate(const uint8_t *in, size_t in_len, uint64_t start_bit,
                 unsigned trailing_bits, uint32_t window_size,
                 const uint8_t *prefix, size_t prefix_len,
                 uint8_t *out, size_t out_cap, orc_result *res);

/* zlib header fields (zlib.lisp:57) */
typedef struct orc_zlib_header {
    uint32_t window_size, level, has_dict, dict_id;
} orc_zlib_header;

/* One zlib member at in[0..in_len): header, deflate, Adler-32 trailer.
 * dict/dict_len: the dictionary the caller's dictionary function returned for
 * hdr->dict_id (NULL = none available -> SEMZ_E_ZLIB_NO_DICT_FN when
 * have_dict_fn==0, SEMZ_E_ZLIB_UNKNOWN_DICT when have_dict_fn!=0). */
void orc_zlib_member(const uint8_t *in, size_t in_len, int have_dict_fn,
                     const uint8_t *dict, size_t dict_len,
                     uint8_t *out, size_t out_cap, orc_zlib_header *hdr, orc_result *res);

/* gzip header fields (gzip.lisp:81-87); strings point into `in` */
typedef struct orc_gzip_header {
    uint32_t textp, mtime, extra_flags, os;
    int64_t  extra_off, extra_len;       /* -1 if absent */
    int64_t  name_off, name_len;
    int64_t  comment_off, comment_len;
    uint64_t header_len;                 /* bytes before the deflate data */
} orc_gzip_header;

void orc_gzip_member(const uint8_t *in, size_t in_len,
                     uint8_t *out, size_t out_cap, orc_gzip_header *hdr, orc_result *res);

/* ---- bzip2 (bzip2.lisp) ---- */
/* One block whose 48-bit magic starts at bit `start_bit`; level = 1..9.
 * On success res->check holds the block CRC (4 bytes, LE of the integer). */
void orc_bzip2_block(const uint8_t *in, size_t in_len, uint64_t start_bit, unsigned level,
                     uint8_t *out, size_t out_cap, orc_result *res);
/* Whole stream ("BZh" header .. footer), one member. res->detail = level. */
void orc_bzip2_member(const uint8_t *in, size_t in_len,
                      uint8_t *out, size_t out_cap, orc_result *res);

/* ---- LZMA2 / XZ (lzma.lisp, lzma2.lisp, xz.lisp, xz-bcj-filters.lisp) ---- */
/* Raw LZMA2 chunk sequence at in[0..limit) (limit = declared compressed size,
 * or in_len when unknown -> have_limit=0). */
/* :raw-lzma (lzma.lisp:809-835); eof_mode 0 :maybe, 1 :always, 2 :never */
void orc_lzma(const uint8_t *in, size_t in_len, unsigned lc, unsigned lp, unsigned pb, uint32_t dict_size,
              int have_size, uint64_t size, int eof_mode, uint8_t *out, size_t out_cap, orc_result *res);
/* :lzma, the .lzma container (lzma.lisp:837-865); honour_size: see lzma.c */
void orc_lzma_alone(const uint8_t *in, size_t in_len, int honour_size, uint8_t *out, size_t out_cap,
                    uint32_t hdr[5], uint64_t *hsize, orc_result *res);
void orc_lzma2(const uint8_t *in, size_t in_len, int have_limit, uint64_t limit,
               uint32_t dict_size, uint8_t *out, size_t out_cap, orc_result *res);
/* x86 BCJ inverse in place over a whole block (xz-bcj-filters.lisp:169-219). */
void orc_bcj_x86(uint8_t *buf, size_t n, uint32_t start_offset);
void orc_bcj_simple(uint8_t *buf, size_t n, uint32_t start_offset, unsigned filter_id);   /* 0x05 .. 0x0A */
/* XZ delta filter inverse in place (xz.lisp:238-258). */
void orc_delta(uint8_t *buf, size_t n, unsigned distance);
/* One xz member (stream header .. footer). res->detail = check type. */
void orc_xz_member(const uint8_t *in, size_t in_len,
                   uint8_t *out, size_t out_cap, orc_result *res);

/* ---- public API mirror (interface.lisp:149-208) ---- */
/* fmt: 0 deflate, 1 zlib, 2 gzip, 3 bzip2, 6 raw-lzma2(dict via opt), 8 xz.
 * all_members != 0 mirrors decompress-all: members are concatenated and
 * trailing data is SEMZ_E_TRAILING_DATA (or the format's own parse error).
 * res->in_end_bit/8 = bytes consumed; res->out_len = total output. */
void orc_decompress(int fmt, const uint8_t *in, size_t in_len, int all_members,
                    uint32_t opt, uint8_t *out, size_t out_cap, orc_result *res);

#ifdef __cplusplus
}
#endif
#endif
