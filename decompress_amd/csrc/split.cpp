#include "semz_host.h"
extern "C" int semz_split(int, const uint8_t *, size_t, const uint64_t *, const uint64_t *, size_t, semz_unit **, size_t *, semz_stream_info *) { return -4; }
