#!/bin/bash
# usage: run_sweep_cw.sh ; runs tests then the perf sweep for both workgroup shapes
set -e
timeout -k 10 240 python -u -m pytest tests/test_gpu_sweep.py -m gpu -x -q --timeout 100 --timeout-method thread > gpurun_out/t6.log 2>&1
SV_SWEEP_CW=15 timeout -k 10 240 python -u -m pytest tests/test_gpu_sweep.py -m gpu -x -q --timeout 100 --timeout-method thread >> gpurun_out/t6.log 2>&1
echo "== CW=7" > gpurun_out/perf_sweep.log
SV_SWEEP_CW=7 timeout -k 10 300 python scratch/perf_sweep.py 1 16 64 256 >> gpurun_out/perf_sweep.log 2>&1
echo "== CW=15" >> gpurun_out/perf_sweep.log
SV_SWEEP_CW=15 timeout -k 10 300 python scratch/perf_sweep.py 1 16 64 256 >> gpurun_out/perf_sweep.log 2>&1
