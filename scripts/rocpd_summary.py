#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db, or the *_kernel_stats.csv /
*_kernel_trace.csv files) into the per-kernel table committed under profiles/."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    return [(n, (e - s)) for n, s, e in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def short_name(n):
    """Drop the trailing argument list and the anonymous-namespace noise of a demangled kernel name."""
    n = n.strip()
    if n.endswith(")"):
        depth = 0
        for i in range(len(n) - 1, -1, -1):
            depth += n[i] == ")"
            depth -= n[i] == "("
            if depth == 0:
                n = n[:i]
                break
    return n.replace("(anonymous namespace)::", "").replace("void ", "").strip()


def main(src, cmd=""):
    rows = []
    if os.path.isdir(src):
        for p in glob.glob(os.path.join(src, "**", "*.db"), recursive=True):
            rows += from_db(p)
        for p in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
            rows += from_csv(p)
    elif src.endswith(".db"):
        rows = from_db(src)
    else:
        rows = from_csv(src)
    agg = {}
    for n, d in rows:
        n = short_name(n)
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary; command: {cmd}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:60]:60s} {a[0]:6d} {a[1]/1e6:12.4f} {a[1]/a[0]/1e6:10.4f} {a[2]/1e6:10.4f} {a[3]/1e6:10.4f} {100*a[1]/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
