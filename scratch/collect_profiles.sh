#!/bin/bash
# usage: collect_profiles.sh <letter>: copies the summaries of gpurun_out/r03_<letter>_* (scratch/prof_all_r03.sh) into profiles/
L=$1
cp gpurun_out/r03_${L}_traffic.json profiles/r03_traffic.json
for c in "c2 1536" "c4 112" "c3 128"; do set -- $c
  cp gpurun_out/r03_${L}_$1/kernel_stats.csv profiles/r03_${L}_kernel_stats_$1_b$2.csv
  cp gpurun_out/r03_${L}_$1/pmc.txt profiles/r03_${L}_pmc_traffic_$1_b$2.txt
done
cp gpurun_out/r03_${L}_bench_c2.json profiles/r03_${L}_bench_c2_b1536.json
cp gpurun_out/r03_${L}_bench_c4.json profiles/r03_${L}_bench_c4_b112.json
cp gpurun_out/r03_${L}_bench_c3.json profiles/r03_${L}_bench_c3_b128.json
cp gpurun_out/r03_${L}_bench_c1.json profiles/r03_${L}_bench_c1_b1024.json
cp gpurun_out/r03_${L}_batches.txt profiles/r03_${L}_throughput_vs_batch_c2.txt
python3 - <<PY
import json
for c in ("c2_b1536", "c4_b112", "c3_b128", "c1_b1024"):
    d = json.loads(open("profiles/r03_${L}_bench_%s.json" % c).read().strip().split("\n")[-1])
    r = d["roofline"]
    print(c, d["value"], d["ms_per_step"], "frac", r.get("frac"), "hbm_actual_frac", r.get("hbm_actual_frac"))
    for k in r.get("kernels", []): print("   ", k["kernel"][:40], k["launch_us"], k["frac"], k.get("hbm_actual_gbs"))
PY
