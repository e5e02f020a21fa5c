#!/bin/bash
# usage: ab_env.sh <tag> <VAR> <value>...   (on the GPU box): stage times of the C2 batch for several values of an environment variable
TAG=$1; VAR=$2; shift; shift
mkdir -p gpurun_out/r3
: > gpurun_out/r3/$TAG.log
for v in "$@"; do
  echo -n "$VAR=$v " >> gpurun_out/r3/$TAG.log
  env $VAR=$v timeout -k 10 200 python scratch/perf_stages.py ${AB_BATCH:-1024} 5 2>&1 | grep fps | sed 's/.*agg/agg/' >> gpurun_out/r3/$TAG.log || exit 1
done
cat gpurun_out/r3/$TAG.log
