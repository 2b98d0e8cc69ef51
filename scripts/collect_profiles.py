#!/usr/bin/env python3
"""Copies the summaries of a profiling session (gpurun_out/<tag>/, made by the rocprofv3 commands quoted in
DESIGN.md section 6) into profiles/: per format the rocprofv3 kernel stats (CSV as written by rocprofv3 plus
a readable table), the bench.py line printed under the profiler, and the PMC traffic entry for gzip.

usage: collect_profiles.py gpurun_out/r01 r01
"""
import csv
import json
import os
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
for fmt in ("gzip", "bzip2", "xz", "xzsha", "zlib"):
    stats = os.path.join(src, f"{fmt}_kt_kernel_stats.csv")
    log = os.path.join(src, f"{fmt}_kt.log")
    if not os.path.exists(stats):
        continue
    rows = list(csv.DictReader(open(stats)))
    line = [l for l in open(log) if l.startswith("{")][-1]
    d = json.loads(line)
    with open(os.path.join(prof, f"{tag}_{fmt}_kernel_stats.txt"), "w") as f:
        args = {"xzsha": "xz --check sha256"}.get(fmt, fmt)
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --format {args} --formats none "
                f"--steps {d['steps']} --warmup {d['warmup']} --no-cpu --no-e2e --workers 1 --corpus-cache DIR   (MI355X; "
                f"scripts/profile_round.sh)\n")
        f.write(f"# workload: {d['config']['workload']}\n")
        f.write(f"{'kernel':34s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}\n")
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
            f.write(f"{name:34s} {int(r['Calls']):6d} {int(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e6:9.4f} "
                    f"{int(r['MinNs']) / 1e6:9.4f} {int(r['MaxNs']) / 1e6:9.4f} {float(r['Percentage']):6.2f}\n")
    with open(os.path.join(prof, f"{tag}_{fmt}_bench_under_rocprof.json"), "w") as f:
        f.write(line)
    print(fmt, d["value"], "GB/s", d["roofline"]["kernel"], d["roofline"]["kernel_ms_avg"], "ms")
fetch, write = os.path.join(src, "gzip_fetch_counter_collection.csv"), os.path.join(src, "gzip_write_counter_collection.csv")
if os.path.exists(fetch) and os.path.exists(write):
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "pmc_traffic.py"), "gzip:4096x262144:inflate_kernel",
                           "inflate_kernel", fetch, write])
