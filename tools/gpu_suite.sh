#!/bin/bash
# GPU: the round's acceptance sequence -- the whole GPU suite, the launches without torch in the process, the driver's
# bench command, and a summary of the line.   usage: bash tools/gpu_suite.sh <tag>   (files under gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-r05}
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=60 > gpurun_out/${tag}_pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/${tag}_pytest_gpu.log
[ $rc -eq 0 ] || exit 1
SF_HIP_NO_TORCH=1 timeout -k 10 200 python tools/no_torch_bench.py > gpurun_out/${tag}_no_torch.log 2>&1; echo "no-torch rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2>gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/${tag}_bench.err
python3 - gpurun_out/${tag}_bench_line.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("value %.4e" % r["value"], "median-based %.4e" % r["value_at_median"], {k: r["roofline"].get(k) for k in ("frac", "basis", "avg_launch_us", "min_us", "median_us", "max_us")})
for o in r.get("other_configs", []):
    if "error" in o: print("ERROR", o); continue
    print("%-60s %.4e (median %.4e) avg launch %.1f us frac %s" % (o["workload"][:60], o["value"], o["value_at_median"], o["roofline"]["avg_launch_us"], o["roofline"].get("frac")))
print(r.get("cpu_baseline"))
PY
