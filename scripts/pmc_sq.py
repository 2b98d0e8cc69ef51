#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: per kernel, the average per launch of every
counter (summed over the dispatch's dimensions), as a plain table for profiles/.

usage: pmc_sq.py OUT.txt TITLE counter_collection.csv [more.csv ...]
"""
import csv
import sys
from collections import defaultdict


def main():
    out, title, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    table = defaultdict(lambda: defaultdict(dict))          # kernel -> counter -> dispatch -> value
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                d = table[k][r["Counter_Name"]]
                d[(path, r["Dispatch_Id"])] = d.get((path, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    with open(out, "w") as f:
        f.write(f"# {title}\n# average per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)\n")
        for k in sorted(table):
            f.write(f"\n{k}\n")
            for c in sorted(table[k]):
                v = table[k][c]
                f.write(f"  {c:28s} {sum(v.values()) / len(v):18.0f}   ({len(v)} launches)\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
