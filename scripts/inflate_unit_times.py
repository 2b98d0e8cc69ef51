#!/usr/bin/env python3
"""Developer diagnostic: when every unit of a gzip bench launch starts and ends, and on which XCD / CU / SIMD it ran
(needs the STAMPS build: make -C decompress_amd/csrc STAMPS=1 OUT=../libsemz_gpu_stamps.so BUILD=build_stamps)."""
import ctypes as C
import os
import sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SEMZ_LIB"] = os.path.join(ROOT, "decompress_amd", "libsemz_gpu_stamps.so")
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"])
import bench
from decompress_amd import _lib
lib = _lib.load()
bench.main()
n = 4096
buf = (C.c_ulonglong * (3 * n))()
assert lib.semz_debug_unit_times(buf, n) == 0
rows = [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]
t0 = min(r[0] for r in rows)
starts = sorted((r[0] - t0) / 100.0 for r in rows)          # microseconds (100 MHz)
ends = sorted((r[1] - t0) / 100.0 for r in rows)
durs = sorted((r[1] - r[0]) / 100.0 for r in rows)
q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
print("starts  us: min %.0f  p50 %.0f  p99 %.0f  max %.0f" % (starts[0], q(starts, .5), q(starts, .99), starts[-1]))
print("ends    us: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (ends[0], q(ends, .1), q(ends, .5), q(ends, .9), q(ends, .99), ends[-1]))
print("lengths us: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f  mean %.0f" % (durs[0], q(durs, .1), q(durs, .5), q(durs, .9), q(durs, .99), durs[-1], sum(durs) / n))
# HW_ID (register 4): wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 ...; XCC_ID register 20
per = defaultdict(list)
for s, e, hw in rows:
    hwid, xcc = hw >> 32, hw & 15
    simd, cu, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 13) & 7
    per[(xcc, se, cu, simd)].append((e - s) / 100.0)
cnt = defaultdict(int)
for k, v in per.items():
    cnt[len(v)] += 1
print("waves per (xcc, se, cu, simd):", dict(sorted(cnt.items())))
by = defaultdict(list)
for k, v in per.items():
    by[len(v)].append(sum(v) / len(v))
for k in sorted(by):
    print("  SIMDs with %d units: mean unit length %.0f us" % (k, sum(by[k]) / len(by[k])))
# does a unit's length follow its data?  (the corpus is rebuilt: same seeds as the bench)
if "--format" not in sys.argv:
    W = bench.build_workload("gzip", n, 262144, 0x5EED0001, 8, "crc64", 1)
    size = [len(b) for b in W["blobs"]]
    d = [(r[1] - r[0]) / 100.0 for r in rows]
    ms, md = sum(size) / n, sum(d) / n
    cov = sum((a - ms) * (b - md) for a, b in zip(size, d))
    corr = cov / (sum((a - ms) ** 2 for a in size) ** .5 * sum((b - md) ** 2 for b in d) ** .5)
    print("compressed bytes per unit: min %d  mean %.0f  max %d;  correlation with the unit's length %.3f" % (min(size), ms, max(size), corr))
    order = sorted(range(n), key=lambda i: size[i])
    for lo in range(0, n, n // 8):
        part = order[lo:lo + n // 8]
        print("  units by size, eighth %d: mean size %.0f  mean length %.0f us" % (lo // (n // 8), sum(size[i] for i in part) / len(part), sum(d[i] for i in part) / len(part)))
# where do the long units run?
for name, key in (("xcc", lambda k: k[0]), ("se", lambda k: k[1]), ("cu", lambda k: k[2]), ("simd", lambda k: k[3])):
    g = defaultdict(list)
    for k, v in per.items():
        g[key(k)] += v
    print("mean unit length by %s:" % name, "  ".join("%s: %.0f" % (a, sum(b) / len(b)) for a, b in sorted(g.items())))
spread = sorted((max(v) - min(v)) for v in per.values())
print("spread of the four units of one SIMD (max - min): p10 %.0f  p50 %.0f  p90 %.0f us" % (q(spread, .1), q(spread, .5), q(spread, .9)))
simd_mean = sorted(sum(v) / len(v) for v in per.values())
print("mean of the four units of one SIMD: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f us" % (simd_mean[0], q(simd_mean, .1), q(simd_mean, .5), q(simd_mean, .9), simd_mean[-1]))
cu = defaultdict(list)
for k, v in per.items():
    cu[k[:3]] += v
cu_mean = sorted(sum(v) / len(v) for v in cu.values())
print("mean of the sixteen units of one CU: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f us" % (cu_mean[0], q(cu_mean, .1), q(cu_mean, .5), q(cu_mean, .9), cu_mean[-1]))
