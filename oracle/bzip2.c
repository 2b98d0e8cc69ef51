#include "oracle.h"
void orc_bzip2_block(const uint8_t *in, size_t in_len, uint64_t start_bit, unsigned level, uint8_t *out, size_t out_cap, orc_result *res){(void)in;(void)in_len;(void)start_bit;(void)level;(void)out;(void)out_cap;res->status=SEMZ_E_UNSUPPORTED;}
void orc_bzip2_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, orc_result *res){(void)in;(void)in_len;(void)out;(void)out_cap;res->status=SEMZ_E_UNSUPPORTED;}
