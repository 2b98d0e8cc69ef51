#!/usr/bin/env python3
"""Folds the counter passes of scripts/profile_counters.sh (gpurun_out/traffic, gpurun_out/sq, gpurun_out/tcc) into
profiles/: pmc_traffic.json entries per format and kernel, <tag>_<fmt>_pmc_sq.txt, <tag>_<fmt>_l2_hit_miss.txt.

usage: collect_counters.py r02
"""
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
KERNELS = {
    "zlib": ["inflate_kernel"],
    "bzip2": ["bz_decode_kernel", "bz_sort_kernel", "bz_walk_kernel", "bz_link_kernel", "bz_rle1_kernel"],
    "xz": ["lzma2_fast_kernel", "bcj_x86_kernel"],
}
for fmt, kernels in KERNELS.items():
    d = os.path.join(out, "traffic", fmt)
    line = os.path.join(d, "bench_line.json")
    if not os.path.exists(line):
        continue
    cfg = json.load(open(line))["config"]
    for k in kernels:
        key = f"{fmt}:{cfg['units_per_gpu']}x{cfg['unit_bytes']}:{k}"
        subprocess.check_call([sys.executable, os.path.join(root, "scripts", "pmc_traffic.py"), key, k,
                               os.path.join(d, "fetch_counter_collection.csv"), os.path.join(d, "write_counter_collection.csv")])
for fmt in ("gzip", "bzip2", "xz"):
    s = os.path.join(out, "sq", fmt, "summary.txt")
    if os.path.exists(s):
        shutil.copy(s, os.path.join(prof, f"{tag}_{fmt}_pmc_sq.txt"))
    t = os.path.join(out, "tcc", f"{fmt}_summary.txt")
    if os.path.exists(t):
        with open(os.path.join(prof, f"{tag}_{fmt}_l2_hit_miss.txt"), "w") as f:
            f.write(f"# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -- python3 bench.py --format {fmt} "
                    "--formats none --steps 2 --warmup 1 --no-cpu --no-e2e --workers 1 --corpus-cache DIR   (scripts/pmc_tcc_run.sh; "
                    "requests, not bytes; EA0_RDREQ = reads the L2 sent on, EA0_RDREQ_DRAM = those that went to HBM, the rest "
                    "were served by the Infinity Cache)\n")
            f.write(open(t).read())
