#!/bin/bash
# One rocprofv3 --pmc pass with L2 (TCC) hit / miss / request counters of `bench.py --format $1`; results in gpurun_out/tcc/$1/.
set -e
F=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tcc/$F
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O -o tcc -- \
    python3 $R/bench.py --format $F --steps 2 --warmup 1 --no-cpu --no-e2e --workers 1 "$@" > $O/tcc.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(o + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"  {c:24s} {v / cnt[(k, c)]:16.0f} per launch ({cnt[(k, c)]} launches)")
PY
