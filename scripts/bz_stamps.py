#!/usr/bin/env python3
"""Developer diagnostic: shader-clock totals and event counts of bz_decode_kernel (needs the STAMPS build:
make -C decompress_amd/csrc STAMPS=1 OUT=../libsemz_gpu_stamps.so BUILD=build_stamps)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SEMZ_LIB"] = os.path.join(ROOT, "decompress_amd", "libsemz_gpu_stamps.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--format", "bzip2", "--steps", "1", "--warmup", "0", "--no-cpu"])
import bench
from decompress_amd import _lib
lib = _lib.load()
bench.main()
st = (C.c_ulonglong * 16)()
assert lib.semz_debug_bz_stamps(st, 1) == 0
b = st[11] or 1
print(f"blocks {b}  Mticks/block: header {st[0] / b / 1e6:.2f}  Huffman wave busy {st[1] / b / 1e6:.2f} waiting {st[3] / b / 1e6:.2f}  "
      f"MTF wave busy {st[2] / b / 1e6:.2f} waiting {st[4] / b / 1e6:.2f}")
print(f"per block: windows {st[5] / b:.0f} symbols {st[6] / b:.0f} windows with long codes {st[7] / b:.0f} stages {st[8] / b:.0f} runs {st[9] / b:.0f}")
print(f"ticks/symbol: Huffman {st[1] / max(1, st[6]):.0f}  MTF {st[2] / max(1, st[6]):.0f}   symbols/window {st[6] / max(1, st[5]):.1f}")
