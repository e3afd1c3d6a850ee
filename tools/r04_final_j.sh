#!/bin/bash
# Round 4, final GPU sequence, part J (radius-3 streaming: the dense skeleton changed once more): fuzz campaigns
# on every generator, the whole GPU suite, rocprofv3 evidence for the workloads whose code objects changed (box, dense),
# then the driver's bench command on the fresh records.
set -o pipefail
OUT=gpurun_out/r04_final_j
mkdir -p $OUT
for g in compact dense dense_sum box_sum; do
  timeout -k 10 100 python tools/star_fuzz.py --generator $g --seeds 400 --first 3000 --seconds 60 > gpurun_out/r04_final5_fuzz_$g.log 2>&1
  echo "fuzz $g rc=$? $(tail -1 gpurun_out/r04_final5_fuzz_$g.log)"
done
SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 100 python tools/star_fuzz.py --generator box_sum --seeds 400 --seconds 60 > gpurun_out/r04_final5_fuzz_box_sum_forced.log 2>&1
echo "fuzz box_sum (dense.t2=2) rc=$? $(tail -1 gpurun_out/r04_final5_fuzz_box_sum_forced.log)"
timeout -k 10 120 python tools/slab_fuzz.py --seeds 200 --seconds 60 > gpurun_out/r04_final5_fuzz_slab.log 2>&1
echo "slab fuzz rc=$? $(tail -1 gpurun_out/r04_final5_fuzz_slab.log)"
SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 120 python tools/slab_fuzz.py --generator box_sum --seeds 200 --seconds 60 > gpurun_out/r04_final5_fuzz_slab_box_sum.log 2>&1
echo "slab fuzz box_sum rc=$? $(tail -1 gpurun_out/r04_final5_fuzz_slab_box_sum.log)"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
SF_PROFILE_WORKLOADS="box dense" SF_PROFILE_SLABS=none bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_7.log 2>&1
echo "profile_round rc=$?"
python3 -c "
import json; t = json.load(open('gpurun_out/hbm_traffic.json'))
for k, v in t.items(): print(k, v.get('workload'), v.get('round'), '%.4f GB' % (v['hbm_bytes_per_launch'] / 1e9), v.get('valu_busy'))"
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
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
