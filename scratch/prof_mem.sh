#!/bin/bash
# usage: prof_mem.sh <tag> [batch]   (on the GPU box, from the repo root): memory-path stall counters (L2 -> fabric write requests,
# L1 / texture-address stalls) over the default bench workload, one rocprofv3 --pmc pass per counter block
set -e
TAG=${1:-r03_mem}; B=${2:-256}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --config c2 --batch $B --steps 2 --warmup 1 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum -d $OUT/p1 -o p -- python3 $ARGS > $OUT/p1.json 2> $OUT/p1.log
timeout -s KILL 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $OUT/p2 -o p -- python3 $ARGS > $OUT/p2.json 2> $OUT/p2.log
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_BUSY_sum -d $OUT/p3 -o p -- python3 $ARGS > $OUT/p3.json 2> $OUT/p3.log
python3 scratch/db_summary.py pmc $OUT/p1/p_results.db $OUT/p2/p_results.db $OUT/p3/p_results.db > $OUT/mem.txt
grep "k_cost2<4, 1024\|k_sweep<4" $OUT/mem.txt
