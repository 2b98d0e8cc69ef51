#include "semz_host.h"
int semz_run_bzip2(semz_ctx *c, const uint8_t *, size_t, const semz_unit *, uint32_t, uint8_t *, size_t, semz_result *, semz_tail *) { return semz_fail(c, -4, "bzip2 not built yet"); }
