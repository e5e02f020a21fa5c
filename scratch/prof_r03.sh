#!/bin/bash
# usage: prof_r03.sh <tag> [config] [batch]   (on the GPU box, from the repo root)
# kernel trace + stats, then the two PMC passes (FETCH_SIZE, WRITE_SIZE), each its own rocprofv3 run.
set -e
TAG=${1:-r03_a}; CFG=${2:-c2}; B=${3:-1536}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --config $CFG --batch $B --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $ARGS > $OUT/stats.json 2> $OUT/stats.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $ARGS > $OUT/write.json 2> $OUT/write.log
python3 scratch/db_summary.py stats $OUT/stats/s_results.db > $OUT/kernel_stats.csv
python3 scratch/db_summary.py pmc $OUT/fetch/f_results.db $OUT/write/w_results.db > $OUT/pmc.txt
python3 scratch/make_traffic.py r03 $CFG $B $OUT/fetch/f_results.db $OUT/write/w_results.db > $OUT/traffic_entry.json
cp profiles/r03_traffic.json $OUT/ 2>/dev/null || true
cat $OUT/kernel_stats.csv
