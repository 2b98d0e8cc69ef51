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
names = ["decode_symbols (symbols decoded and applied)", "wait for a far match's bytes (C++ resolve)", "after a run: checks", "flush",
         "C++ match: resolve, sync", "C++ match: copy", "unit end", "loop top / other"]
units = st[14] or 1
tot = sum(st[:8]) or 1
for i, n in enumerate(names):
    print(f"{n:48s} {st[i] / units / 1e6:10.3f} Mticks/unit {100 * st[i] / tot:6.1f}%")
runs_lit, ring, to_cpp, runs_match, far_long = (st[8 + i] / units for i in range(5))
print(f"per unit: runs ended behind a literal {runs_lit:.0f}, behind a match {runs_match:.0f}, by a match for the C++ code {to_cpp:.0f} "
      f"(ring copies {ring:.0f}, long far copies {far_long:.0f})")
print(f"ticks per unit: {tot / units / 1e6:.1f} M; runs per unit: {runs_lit + runs_match + to_cpp:.0f}")
why = st[15]
print("units handed to the general kernel, by reason (1 direct bits, 2 end marker / input beyond the chunk, 3 dict size, 4 no data, "
      "5 beyond the chunk):", [(why >> (8 * i)) & 0xFF for i in range(1, 6)], "(counts modulo 256)")
