#include "oracle.h"
void orc_bcj_x86(uint8_t *buf, size_t n, uint32_t start_offset){(void)buf;(void)n;(void)start_offset;}
void orc_delta(uint8_t *buf, size_t n, unsigned distance){(void)buf;(void)n;(void)distance;}
void orc_xz_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, orc_result *res){(void)in;(void)in_len;(void)out;(void)out_cap;res->status=SEMZ_E_UNSUPPORTED;}
void orc_xz_next_member_start(const uint8_t *in, size_t in_len, size_t pos, size_t *next, int *status){(void)in;(void)in_len;(void)pos;(void)next;*status=SEMZ_E_UNSUPPORTED;}
