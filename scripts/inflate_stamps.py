#!/usr/bin/env python3
"""Developer diagnostic: per-phase shader-clock totals of inflate_kernel (needs the STAMPS build:
make -C decompress_amd/csrc STAMPS=1 OUT=../libsemz_gpu_stamps.so BUILD=build_stamps)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SEMZ_LIB"] = os.path.join(ROOT, "decompress_amd", "libsemz_gpu_stamps.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--steps", "1", "--warmup", "0", "--no-cpu"])
import bench
from decompress_amd import _lib
lib = _lib.load()
bench.main()
st = (C.c_ulonglong * 16)()
assert lib.semz_debug_stamps(st, 1) == 0
names = ["loop-top", "decode", "classify", "far", "raw+near", "flush", "tables", "-"]
units = st[12] or 1
tot = sum(st[:8]) or 1
for i, n in enumerate(names):
    print(f"{n:10s} {st[i] / units / 1e6:10.3f} Mcycles/unit {100 * st[i] / tot:6.1f}%")
print(f"tokens/unit {st[8] / units:.0f} rounds/unit {st[9] / units:.0f} far/unit {st[10] / units:.0f} near/unit {st[11] / units:.0f} units {units}")
print(f"cycles/token total {tot / max(1, st[8]):.0f}  decode only {st[1] / max(1, st[8]):.0f}")
