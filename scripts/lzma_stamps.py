#!/usr/bin/env python3
"""Developer diagnostic: where the wave time of lzma2_fast_kernel goes (needs the STAMPS build:
make -C decompress_amd/csrc STAMPS=1 OUT=../libsemz_gpu_stamps.so BUILD=build_stamps)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SEMZ_LIB"] = os.path.join(ROOT, "decompress_amd", "libsemz_gpu_stamps.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--format", "xz", "--units", "16", "--replicate", "8", "--steps", "1", "--warmup", "0",
                                             "--no-cpu", "--no-e2e", "--formats", "none"])
import bench
from decompress_amd import _lib
lib = _lib.load()
bench.main()
st = (C.c_ulonglong * 16)()
assert lib.semz_debug_lz_stamps(st, 1) == 0
names = ["decode_symbol", "wait for a far match's bytes", "literal: apply", "flush", "match: checks, resolve, sync", "match: copy",
         "unit end", "loop top / other"]
units = st[14] or 1
tot = sum(st[:8]) or 1
for i, n in enumerate(names):
    print(f"{n:32s} {st[i] / units / 1e6:10.3f} Mticks/unit {100 * st[i] / tot:6.1f}%")
lit, mlit, match, far, _slow = (st[8 + i] / units for i in range(5))
print(f"per unit: literals {lit:.0f} literals behind a match {mlit:.0f} matches {match:.0f} of them far (deferred) {far:.0f} long far {_slow:.0f}")
print(f"ticks per symbol: {tot / units / max(1, lit + mlit + match):.0f}; decode_symbol alone {st[0] / units / max(1, lit + mlit + match):.0f}")
why = st[15]
print("units handed to the general kernel, by reason (1 direct bits, 2 end marker, 3 dict size, 4 no data, 5 beyond the chunk):",
      [(why >> (8 * i)) & 0xFF for i in range(1, 6)], "(counts modulo 256)")
