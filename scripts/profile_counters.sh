#!/bin/bash
# The round's counter passes, run on the GPU box through gpurun after scripts/profile_round.sh:
#   gpurun -- 'bash scripts/profile_counters.sh'
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) for zlib, bzip2 and xz (gzip's are part of profile_round.sh),
# the SQ instruction / wait / LDS counters for gzip, bzip2 and xz, and the L2 hit / miss counters for gzip and bzip2.
# scripts/collect_counters.py folds the results into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
XZ="--units 16 --replicate 8"
bash $R/scripts/pmc_traffic_run.sh zlib > /dev/null && echo "traffic zlib done" || exit 1
bash $R/scripts/pmc_traffic_run.sh bzip2 > /dev/null && echo "traffic bzip2 done" || exit 1
bash $R/scripts/pmc_traffic_run.sh xz $XZ > /dev/null && echo "traffic xz done" || exit 1
bash $R/scripts/pmc_sq_run.sh gzip > /dev/null && echo "sq gzip done" || exit 1
bash $R/scripts/pmc_sq_run.sh bzip2 > /dev/null && echo "sq bzip2 done" || exit 1
bash $R/scripts/pmc_sq_run.sh xz $XZ > /dev/null && echo "sq xz done" || exit 1
mkdir -p $R/gpurun_out/tcc
bash $R/scripts/pmc_tcc_run.sh gzip > $R/gpurun_out/tcc/gzip_summary.txt && echo "tcc gzip done" || exit 1
bash $R/scripts/pmc_tcc_run.sh bzip2 > $R/gpurun_out/tcc/bzip2_summary.txt && echo "tcc bzip2 done" || exit 1
