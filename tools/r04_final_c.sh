#!/bin/bash
# Round 4, final GPU sequence, part C: fuzz campaign on the final code (plan-time self-check on), every kernel family,
# copy boundaries, slab decomposition; then the bench line with the freshly committed PMC records.
set -o pipefail
OUT=gpurun_out/r04_final_fuzz
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? : $(tail -1 $OUT/$name.log)"; }
run star python tools/star_fuzz.py --first 5000 --seeds 2000 --seconds 100
run dag python tools/star_fuzz.py --generator dag --first 5000 --seeds 2000 --seconds 120
run dag_w5 python tools/star_fuzz.py --generator dag --first 7000 --seeds 2000 --seconds 70 --options "dag.windows=5;fuse=3"
run wide python tools/star_fuzz.py --generator wide --first 5000 --seeds 2000 --seconds 60
run compact python tools/star_fuzz.py --generator compact --first 5000 --seeds 2000 --seconds 60
run dense python tools/star_fuzz.py --generator dense --first 5000 --seeds 2000 --seconds 45
run dense_sum python tools/star_fuzz.py --generator dense_sum --first 5000 --seeds 2000 --seconds 30
run copy python tools/star_fuzz.py --copy --first 5000 --seeds 2000 --seconds 50
run slab python tools/slab_fuzz.py --first 5000 --seeds 400 --seconds 70
run slab_dag python tools/slab_fuzz.py --generator dag --first 5000 --seeds 400 --seconds 60
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2>$OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_final_fuzz/bench_line.json"))
print("value %.4e" % r["value"], {k: r["roofline"].get(k) for k in ("frac", "basis", "avg_launch_us", "min_us", "median_us", "max_us", "valu_waves_active_per_simd")})
for o in r["other_configs"]:
    if "error" in o: print("ERROR", o); continue
    ro = o["roofline"]
    print("%-56s %.4e frac %.3f basis %s busy %s %s" % (o["workload"][:56], o["value"], ro["frac"], ro["basis"], ro.get("valu_waves_active_per_simd"), (ro.get("program") or {}).get("pmc_over_compulsory")))
PY
