#include "semz_host.h"
extern "C" int semz_decode_stream(semz_ctx *, int, const uint8_t *, size_t, int, uint8_t **, size_t *, semz_result *, semz_stream_info *) { return -4; }
