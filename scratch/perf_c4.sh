#!/bin/bash
# usage: perf_c4.sh <tag>   (on the GPU box): C4 bench line with and without the disparity parts of the cost kernel
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/$1_c4.json 2> gpurun_out/r3/$1_c4.err || exit 1
SV_COST_NOPARTS=1 timeout -k 10 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/$1_c4_noparts.json 2> gpurun_out/r3/$1_c4_noparts.err || exit 1
python3 - <<PY
import json
for f in ("gpurun_out/r3/$1_c4.json", "gpurun_out/r3/$1_c4_noparts.json"):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], [(k["kernel"][:28], round(k["launch_us"] / 1e3, 2)) for k in d["roofline"]["kernels"]])
PY
