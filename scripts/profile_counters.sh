#!/bin/bash
# The round's counter passes, run on the GPU box through gpurun after scripts/profile_round.sh:
#   gpurun --timeout 1200 -- 'bash scripts/profile_counters.sh'
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) for zlib, bzip2 and xz (gzip's are part of profile_round.sh),
# the SQ instruction / wait / LDS counters for gzip, bzip2 and xz, and the L2 hit / miss / DRAM-read counters for
# gzip and bzip2.  scripts/collect_counters.py folds the results into profiles/.  Corpora: see profile_round.sh.
R=${GRAFT_REPO_ROOT:-/root/repo}
C=/tmp/semz_corpus
mkdir -p $C $R/gpurun_out/tcc
for F in gzip zlib bzip2 xz; do
    timeout -k 10 300 python3 $R/bench.py --format $F --formats none --make-corpus --corpus-cache $C > /dev/null 2>&1 || { echo "$F: corpus failed"; exit 1; }
    echo "$F corpus done"
done
X="--formats none --corpus-cache $C"
bash $R/scripts/pmc_traffic_run.sh zlib $X > /dev/null && echo "traffic zlib done" || exit 1
bash $R/scripts/pmc_traffic_run.sh bzip2 $X > /dev/null && echo "traffic bzip2 done" || exit 1
bash $R/scripts/pmc_traffic_run.sh xz $X > /dev/null && echo "traffic xz done" || exit 1
bash $R/scripts/pmc_sq_run.sh gzip $X > /dev/null && echo "sq gzip done" || exit 1
bash $R/scripts/pmc_sq_run.sh bzip2 $X > /dev/null && echo "sq bzip2 done" || exit 1
bash $R/scripts/pmc_sq_run.sh xz $X > /dev/null && echo "sq xz done" || exit 1
bash $R/scripts/pmc_tcc_run.sh gzip $X > $R/gpurun_out/tcc/gzip_summary.txt && echo "tcc gzip done" || exit 1
bash $R/scripts/pmc_tcc_run.sh bzip2 $X > $R/gpurun_out/tcc/bzip2_summary.txt && echo "tcc bzip2 done" || exit 1
