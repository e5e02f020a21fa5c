#!/bin/bash
# usage: prof_sq.sh <tag> [batch]   (on the GPU box, from the repo root): SQ / GRBM counter passes over the default bench workload
set -e
TAG=${1:-r02_sq}; B=${2:-256}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --config c2 --batch $B --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- python3 $ARGS > $OUT/p1.json 2> $OUT/p1.log
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/p2 -o p -- python3 $ARGS > $OUT/p2.json 2> $OUT/p2.log
python3 scratch/db_summary.py pmc $OUT/p1/p_results.db $OUT/p2/p_results.db > $OUT/sq.txt
cat $OUT/sq.txt
