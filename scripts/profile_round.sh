#!/bin/bash
# The round's profiling session, run on the GPU box through gpurun:
#   gpurun --timeout 1200 -- 'bash scripts/profile_round.sh r03'
# Per format: rocprofv3 --kernel-trace --stats of bench.py; for gzip also the two HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  scripts/collect_profiles.py copies the summaries to profiles/.
# The corpora are generated first, outside the profiler (bench.py --make-corpus): a process under rocprofv3 starts
# no children, and the xz workload is the default one (64 distinct streams, each twice).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
C=/tmp/semz_corpus
mkdir -p $O $C
cd /tmp && export TMPDIR=/tmp
args() {
    case $1 in
        xz) echo "--format xz" ;;
        xzsha) echo "--format xz --check sha256" ;;
        *) echo "--format $1" ;;
    esac
}
for F in gzip zlib bzip2 xz xzsha; do
    timeout -k 10 300 python3 $R/bench.py $(args $F) --formats none --make-corpus --corpus-cache $C > $O/${F}_corpus.log 2>&1 || { echo "$F: corpus failed"; tail -3 $O/${F}_corpus.log; exit 1; }
    echo "$F corpus done"
done
for F in gzip zlib bzip2 xz xzsha; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${F}_kt -- \
        python3 $R/bench.py $(args $F) --formats none --steps 5 --warmup 2 --no-cpu --no-e2e --workers 1 --corpus-cache $C > $O/${F}_kt.log 2>&1 \
        || { echo "$F: profiled run failed"; tail -3 $O/${F}_kt.log; exit 1; }
    find $O -name "${F}_kt_kernel_stats.csv" -exec cp {} $O/ \; 2>/dev/null || true
    echo "$F done"
done
for CN in FETCH_SIZE WRITE_SIZE; do
    N=$(echo $CN | tr A-Z a-z | cut -d_ -f1)
    timeout -s KILL 170 rocprofv3 --pmc $CN --output-format csv -d $O -o gzip_$N -- \
        python3 $R/bench.py --format gzip --formats none --steps 2 --warmup 1 --no-cpu --no-e2e --workers 1 --corpus-cache $C > $O/gzip_$N.log 2>&1 \
        || { echo "$CN: pass failed"; tail -3 $O/gzip_$N.log; exit 1; }
    find $O -name "gzip_${N}_counter_collection.csv" -exec cp {} $O/ \; 2>/dev/null || true
    echo "$CN done"
done
ls $O
