#!/bin/bash
# The round's profiling session, run on the GPU box through gpurun:
#   gpurun -- 'bash scripts/profile_round.sh r01'
# Per format: rocprofv3 --kernel-trace --stats of bench.py; for gzip also the two HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  scripts/collect_profiles.py copies the summaries to profiles/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (xz: 16 distinct streams, each eight times, so that one process generates the corpus in the time allowed; the
# block count and shape are those of the default run.  xzsha: the same with SHA-256 as the block check, BASELINE configs[3].)
for F in gzip zlib bzip2 xz xzsha; do
    case $F in
        xz) ARGS="--format xz --units 16 --replicate 8" ;;
        xzsha) ARGS="--format xz --check sha256 --units 16 --replicate 8" ;;
        *) ARGS="--format $F" ;;
    esac
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${F}_kt -- \
        python3 $R/bench.py $ARGS --steps 5 --warmup 2 --no-cpu --no-e2e --workers 1 > $O/${F}_kt.log 2>&1
    find $O -name "${F}_kt_kernel_stats.csv" -exec cp {} $O/ \; 2>/dev/null || true
    echo "$F done"
done
for C in FETCH_SIZE WRITE_SIZE; do
    N=$(echo $C | tr A-Z a-z | cut -d_ -f1)
    timeout -s KILL 170 rocprofv3 --pmc $C --output-format csv -d $O -o gzip_$N -- \
        python3 $R/bench.py --format gzip --steps 2 --warmup 1 --no-cpu --no-e2e --workers 1 > $O/gzip_$N.log 2>&1
    find $O -name "gzip_${N}_counter_collection.csv" -exec cp {} $O/ \; 2>/dev/null || true
    echo "$C done"
done
ls $O
