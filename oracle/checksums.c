/*
 * checksums.c -- oracle restatement of the reference's checksums.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   CRC-32   crc.lisp:19-110   reflected 0xEDB88320, init ~0, xorout ~0
 *   CRC-64   crc.lisp:122-193  reflected 0xC96C5795D7870F42, init ~0, xorout ~0
 *   Adler-32 adler32.lisp:54-87
 *   bzip2 CRC bzip2.lisp:13-39 (poly 0x04C11DB7 MSB-first; tables.lisp:10-82 is
 *            the table this polynomial generates)
 *   SHA-256  sha256.lisp:27-164 (RFC 6234)
 */
#include "oracle.h"
#include <string.h>

static uint32_t crc32_tab[256];
static uint64_t crc64_tab[256];
static uint32_t bzcrc_tab[256];
static int tabs_ready;

static void init_tabs(void)
{
    if (tabs_ready) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;                         /* crc.lisp:24-30 */
        for (int j = 0; j < 8; j++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        crc32_tab[i] = c;
        uint64_t d = i;                         /* crc.lisp:127-133 */
        for (int j = 0; j < 8; j++) d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
        crc64_tab[i] = d;
        uint32_t b = i << 24;                   /* bzip2 refimpl table, tables.lisp:10-82 */
        for (int j = 0; j < 8; j++) b = (b & 0x80000000u) ? (b << 1) ^ 0x04C11DB7u : b << 1;
        bzcrc_tab[i] = b;
    }
    tabs_ready = 1;
}

uint32_t orc_crc32_update(uint32_t s, const uint8_t *p, size_t n)
{
    init_tabs();
    for (size_t i = 0; i < n; i++)              /* crc.lisp:43-47 */
        s = crc32_tab[(s ^ p[i]) & 0xFF] ^ (s >> 8);
    return s;
}
uint32_t orc_crc32(const uint8_t *p, size_t n) { return ~orc_crc32_update(0xFFFFFFFFu, p, n); }

uint64_t orc_crc64_update(uint64_t s, const uint8_t *p, size_t n)
{
    init_tabs();
    for (size_t i = 0; i < n; i++)              /* crc.lisp:159-162 */
        s = crc64_tab[(s & 0xFF) ^ p[i]] ^ (s >> 8);
    return s;
}
uint64_t orc_crc64(const uint8_t *p, size_t n) { return ~orc_crc64_update(~0ull, p, n); }

uint32_t orc_adler32_update(uint32_t state, const uint8_t *p, size_t n)
{
    uint32_t s1 = state & 0xFFFF, s2 = state >> 16;   /* adler32.lisp:60-61 */
    while (n) {
        size_t k = n < 5552 ? n : 5552;               /* deferred modulo, adler32.lisp:63-70 */
        for (size_t i = 0; i < k; i++) { s1 += p[i]; s2 += s1; }
        s1 %= 65521; s2 %= 65521;
        p += k; n -= k;
    }
    return (s2 << 16) | s1;
}
uint32_t orc_adler32(const uint8_t *p, size_t n) { return orc_adler32_update(1, p, n); }

/* The reference byte-swaps the table and the state so that it can reuse the
 * reflected CRC-32 loop (bzip2.lisp:13-37); the value it compares is the plain
 * MSB-first CRC, which is what is computed here. */
uint32_t orc_bzcrc_update(uint32_t s, const uint8_t *p, size_t n)
{
    init_tabs();
    for (size_t i = 0; i < n; i++)
        s = (s << 8) ^ bzcrc_tab[(s >> 24) ^ p[i]];
    return s;
}
uint32_t orc_bzcrc(const uint8_t *p, size_t n) { return ~orc_bzcrc_update(0xFFFFFFFFu, p, n); }

/* ---- SHA-256, RFC 6234 (sha256.lisp:27-99; constants tables.lisp:152-197) ---- */
static const uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2
};
#define ROTR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_block(uint32_t h[8], const uint8_t *b)
{
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)b[4*i] << 24 | (uint32_t)b[4*i+1] << 16 | (uint32_t)b[4*i+2] << 8 | b[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROTR(w[i-15], 7) ^ ROTR(w[i-15], 18) ^ (w[i-15] >> 3);
        uint32_t s1 = ROTR(w[i-2], 17) ^ ROTR(w[i-2], 19) ^ (w[i-2] >> 10);
        w[i] = s1 + w[i-7] + s0 + w[i-16];
    }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ROTR(e, 6) ^ ROTR(e, 11) ^ ROTR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROTR(a, 2) ^ ROTR(a, 13) ^ ROTR(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

/* digest[] is the standard big-endian-word digest; read as a little-endian
 * integer it equals finish-sha-256's return value (sha256.lisp:159-164). */
void orc_sha256(const uint8_t *p, size_t n, uint8_t digest[32])
{
    uint32_t h[8] = { 0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19 };
    size_t i = 0;
    for (; i + 64 <= n; i += 64) sha256_block(h, p + i);
    uint8_t tail[128];
    size_t r = n - i;
    memset(tail, 0, sizeof tail);
    memcpy(tail, p + i, r);
    tail[r] = 0x80;
    size_t tl = (r + 1 + 8 <= 64) ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; k++) {
        digest[4*k] = h[k] >> 24; digest[4*k+1] = h[k] >> 16; digest[4*k+2] = h[k] >> 8; digest[4*k+3] = h[k];
    }
}
