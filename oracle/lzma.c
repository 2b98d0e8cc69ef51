#include "oracle.h"
void orc_lzma2(const uint8_t *in, size_t in_len, int have_limit, uint64_t limit, uint32_t dict_size, uint8_t *out, size_t out_cap, orc_result *res){(void)in;(void)in_len;(void)have_limit;(void)limit;(void)dict_size;(void)out;(void)out_cap;res->status=SEMZ_E_UNSUPPORTED;}
