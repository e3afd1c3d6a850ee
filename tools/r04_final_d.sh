#!/bin/bash
# Round 4, final GPU sequence, part D: the whole GPU suite on the final tree, then the driver's bench command.
set -o pipefail
OUT=gpurun_out/r04_final_d
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2>$OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_final_d/bench_line.json"))
print("value %.4e" % r["value"], {k: r["roofline"].get(k) for k in ("frac", "basis", "avg_launch_us", "min_us", "median_us", "max_us")})
for o in r["other_configs"]:
    if "error" in o: print("ERROR", o); continue
    ro = o["roofline"]
    print("%-56s %.4e frac %.3f basis %s %s" % (o["workload"][:56], o["value"], ro["frac"], ro["basis"], (ro.get("program") or {}).get("pmc_over_compulsory")))
PY
