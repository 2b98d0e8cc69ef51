#!/usr/bin/env python3
"""Developer diagnostic: per-phase shader-clock totals of inflate_kernel (needs the STAMPS build:
make -C decompress_amd/csrc STAMPS=1 OUT=../libsemz_gpu_stamps.so BUILD=build_stamps)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SEMZ_LIB"] = os.environ.get("SEMZ_STAMPS_LIB") or os.path.join(ROOT, "decompress_amd", "libsemz_gpu_stamps.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--steps", "1", "--warmup", "0", "--no-cpu"])
import bench
from decompress_amd import _lib
lib = _lib.load()
bench.main()
st = (C.c_ulonglong * 16)()
assert lib.semz_debug_stamps(st, 1) == 0
names = ["decode", "(unused)", "resolve", "flush", "tables+end", "loop-top"]
units = st[14] or 1
tot = sum(st[:6]) or 1
for i, n in enumerate(names):
    if n != "(unused)":
        print(f"{n:10s} {st[i] / units / 1e6:10.3f} Mticks/unit {100 * st[i] / tot:6.1f}%")
win, tok, careful, rounds, far, near = (st[6 + i] / units for i in range(6))
print(f"per unit: windows {win:.0f} window tokens {tok:.0f} ({tok / max(1, win):.2f} per window) scalar steps {careful:.0f} rounds {rounds:.0f} "
      f"matches resolved together (source in front of the round) {far:.0f}, one after the other (source inside the round) {near:.0f}")
print(f"windows not kept whole: {st[12] / units:.0f} per unit, of them for the scalar step {st[13] / units:.0f}")
print(f"ticks per token: total {tot / units / max(1, tok + careful):.0f}  decode only {st[0] / units / max(1, tok + careful):.0f}")
