#!/bin/bash
# usage: prof_all_r03.sh <letter>   (on the GPU box): the round's profile set from the sources in the tree
# kernel stats + FETCH/WRITE PMC for C2 (B=1536), C4 (B=98), C3 (B=128); SQ counters for C2 and C4; throughput vs batch
L=${1:-d}
rm -f profiles/r03_traffic.json
for spec in "c2 1536" "c4 112" "c3 128"; do
  set -- $spec
  scratch/prof_r03.sh r03_${L}_$1 $1 $2 > gpurun_out/r03_${L}_$1.log 2>&1 || { echo "profile $1 failed"; tail -5 gpurun_out/r03_${L}_$1.log; exit 1; }
  rm -rf gpurun_out/r03_${L}_$1/stats gpurun_out/r03_${L}_$1/fetch gpurun_out/r03_${L}_$1/write
done
cp profiles/r03_traffic.json gpurun_out/r03_${L}_traffic.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r03_${L}_bench_c2.json 2> gpurun_out/r03_${L}_bench_c2.err || exit 1
timeout -k 10 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_${L}_bench_c4.json 2> gpurun_out/r03_${L}_bench_c4.err || exit 1
timeout -k 10 300 python bench.py --config c3 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_${L}_bench_c3.json 2> gpurun_out/r03_${L}_bench_c3.err || exit 1
timeout -k 10 300 python bench.py --config c1 --batch 1024 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_${L}_bench_c1.json 2> gpurun_out/r03_${L}_bench_c1.err || exit 1
timeout -k 10 300 python scratch/perf_batches.py > gpurun_out/r03_${L}_batches.txt 2>&1 || exit 1
echo done
