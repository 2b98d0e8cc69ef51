#!/bin/bash
# Two rocprofv3 --pmc passes (instruction mix; cycles, waits and LDS conflicts) of `bench.py --format $1`,
# 8 SQ counters each (the per-pass limit).  Run on the GPU box through gpurun; results in gpurun_out/sq/$1/.
set -e
F=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq/$F
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH \
    -d $O/p1 -o p1 --output-format csv -- python3 $R/bench.py --format $F --steps 2 --warmup 1 --workers 1 --no-cpu --no-e2e "$@" > $O/p1.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    -d $O/p2 -o p2 --output-format csv -- python3 $R/bench.py --format $F --steps 2 --warmup 1 --workers 1 --no-cpu --no-e2e "$@" > $O/p2.log 2>&1
python3 $R/scripts/pmc_sq.py $O/summary.txt "bench.py --format $F $* (2 timed steps + 1 warmup per pass)" $(find $O -name "*counter_collection.csv")
