#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes (they do not fit one pass) of `bench.py --format $1`; results in gpurun_out/traffic/$1/.
set -e
F=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/traffic/$F
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    N=$(echo $C | tr A-Z a-z | cut -d_ -f1)
    timeout -s KILL 170 rocprofv3 --pmc $C --output-format csv -d $O -o $N -- \
        python3 $R/bench.py --format $F --steps 2 --warmup 1 --no-cpu --no-e2e --workers 1 "$@" > $O/$N.log 2>&1
    find $O -name "${N}_counter_collection.csv" -exec cp {} $O/ \; 2>/dev/null || true
done
grep "^{" $O/fetch.log | tail -1 > $O/bench_line.json
ls $O
