#!/bin/bash
# usage: ab.sh <tag> <lib suffix>...   (on the GPU box): stage times of the C2 batch for several builds of the library
TAG=$1; shift
mkdir -p gpurun_out/r3
: > gpurun_out/r3/$TAG.log
for v in "$@"; do
  lib=$PWD/stereo-vision_amd/libsv_hip${v}.so
  SVHIP_LIB=$lib timeout -k 10 200 python scratch/perf_stages.py ${AB_BATCH:-1024} 5 2>&1 | grep fps >> gpurun_out/r3/$TAG.log || exit 1
done
cat gpurun_out/r3/$TAG.log
