#!/bin/bash
# Round 4, final GPU sequence, part A: the whole GPU suite, the launches without torch in the process, the bench line.
set -o pipefail
OUT=gpurun_out/r04_final
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
SF_HIP_NO_TORCH=1 timeout -k 10 200 python tools/no_torch_bench.py > $OUT/no_torch.log 2>&1; echo "no-torch rc=$?"; grep "^{" $OUT/no_torch.log | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2>$OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_final/bench_line.json"))
print("value %.4e" % r["value"], "median-based %.4e" % r["value_at_median"], {k: r["roofline"].get(k) for k in ("frac", "basis", "avg_launch_us", "min_us", "median_us", "max_us")})
for o in r["other_configs"]:
    if "error" in o: print("ERROR", o); continue
    print("%-60s %.4e (median %.4e) avg launch %.1f us" % (o["workload"][:60], o["value"], o["value_at_median"], o["roofline"]["avg_launch_us"]))
print(r["cpu_baseline"])
PY
