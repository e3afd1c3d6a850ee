#!/bin/bash
# Round 4, final GPU sequence, part E: the star kernels were re-hashed once more (codegen text) -- their rocprofv3
# evidence again (c3 c2 c5 fork + slab kernels), the tile-copy ceiling, then the bench line on the fresh records.
SF_PROFILE_WORKLOADS="c3 c2 c5 fork" SF_PROFILE_SLABS="2 4 8" bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_3.log 2>&1
echo "profile_round rc=$?"
python3 -c "
import json; t = json.load(open('gpurun_out/hbm_traffic.json'))
for k, v in t.items(): print(k, v.get('workload'), v.get('round'), '%.4f GB' % (v['hbm_bytes_per_launch'] / 1e9), v.get('valu_busy'))"
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
./tools/micro/tile_copy > gpurun_out/r04_tile_copy.log 2>&1; echo "tile_copy rc=$?"; cat gpurun_out/r04_tile_copy.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2>gpurun_out/r04_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/r04_bench_line.json"))
print("value %.4e" % r["value"], {k: r["roofline"].get(k) for k in ("frac", "basis", "avg_launch_us", "min_us", "median_us", "max_us")})
for o in r["other_configs"]:
    if "error" in o: print("ERROR", o); continue
    ro = o["roofline"]
    print("%-56s %.4e frac %.3f basis %s %s" % (o["workload"][:56], o["value"], ro["frac"], ro["basis"], (ro.get("program") or {}).get("pmc_over_compulsory")))
PY
