#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; TCC has four counter
slots and the pair needs five) into the per-launch HBM traffic entry of profiles/pmc_traffic.json.

usage: pmc_traffic.py KEY KERNEL_SUBSTRING fetch_counter_collection.csv write_counter_collection.csv [out.json]

Units and corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE reports one half of the bytes of wide coalesced reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Narrower accesses are uncalibrated, which the entry says.
"""
import csv
import json
import os
import sys


def per_launch(path, kernel, counter):
    per_dispatch = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per_dispatch[r["Dispatch_Id"]] = per_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    if not per_dispatch:
        raise SystemExit(f"{path}: no {counter} rows for a kernel matching {kernel!r}")
    return sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch)


def main():
    key, kernel, fetch_csv, write_csv = sys.argv[1:5]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "pmc_traffic.json")
    fetch_kib, nf = per_launch(fetch_csv, kernel, "FETCH_SIZE")
    write_kib, nw = per_launch(write_csv, kernel, "WRITE_SIZE")
    entry = {
        "kernel": kernel, "launches_fetch_pass": nf, "launches_write_pass": nw,
        "FETCH_SIZE_KiB_per_launch": round(fetch_kib, 1), "WRITE_SIZE_KiB_per_launch": round(write_kib, 1),
        "read_bytes_per_launch": int(2 * 1024 * fetch_kib), "write_bytes_per_launch": int(1024 * write_kib),
        "traffic_bytes_per_launch": int(2 * 1024 * fetch_kib + 1024 * write_kib),
        "correction": "FETCH_SIZE x 1024 x 2 (gfx950 counts 128-byte requests as 64), WRITE_SIZE x 1024; "
                      "exact for 16-byte-per-lane accesses, uncalibrated for narrower ones",
    }
    table = {}
    if os.path.exists(out):
        with open(out) as f:
            table = json.load(f)
    table[key] = entry
    with open(out, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
        f.write("\n")
    print(key, json.dumps(entry))


if __name__ == "__main__":
    main()
