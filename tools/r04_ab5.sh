#!/bin/bash
# Round 4, GPU session 5: wave -> SIMD placement probe; C5 with / without the staging registers (k1.pf2) under both
# device compilers (PyTorch's comgr, ROCm 7.2's through SF_HIP_COMGR); C3 under both; the new bench line; the suite.
set -o pipefail
OUT=gpurun_out/r04_ab5
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab5
./tools/micro/simd_map > $OUT/simd_map.log 2>&1; echo "simd_map rc=$?"; cat $OUT/simd_map.log
B5="python bench.py --workload c5 --stages 300 --steps 10 --warmup 2"
B3="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
show() { python -c "
import json,sys; r = json.load(open(sys.argv[1]))
print('%-46s' % sys.argv[2], '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][:100], '|', r['config'].get('compiler','')[-60:])" $1 "$2"; }
for round in 1 2; do
  echo "== round $round"
  $B5 > $OUT/c5_torch_default_$round.json 2>$OUT/err.log && show $OUT/c5_torch_default_$round.json "c5 torch-comgr default" || tail -3 $OUT/err.log
  $B5 --options "k1.pf2=0" > $OUT/c5_torch_pf20_$round.json 2>$OUT/err.log && show $OUT/c5_torch_pf20_$round.json "c5 torch-comgr k1.pf2=0" || tail -3 $OUT/err.log
  SF_HIP_COMGR=/opt/rocm/lib/libamd_comgr.so.3 $B5 > $OUT/c5_rocm_default_$round.json 2>$OUT/err.log && show $OUT/c5_rocm_default_$round.json "c5 rocm-comgr default" || tail -3 $OUT/err.log
  SF_HIP_COMGR=/opt/rocm/lib/libamd_comgr.so.3 $B5 --options "k1.pf2=0" > $OUT/c5_rocm_pf20_$round.json 2>$OUT/err.log && show $OUT/c5_rocm_pf20_$round.json "c5 rocm-comgr k1.pf2=0" || tail -3 $OUT/err.log
  $B3 > $OUT/c3_torch_$round.json 2>$OUT/err.log && show $OUT/c3_torch_$round.json "c3 torch-comgr" || tail -3 $OUT/err.log
  SF_HIP_COMGR=/opt/rocm/lib/libamd_comgr.so.3 $B3 > $OUT/c3_rocm_$round.json 2>$OUT/err.log && show $OUT/c3_rocm_$round.json "c3 rocm-comgr" || tail -3 $OUT/err.log
done
echo "== bench.py (driver's command line)"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2>$OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_ab5/bench_line.json"))
print("value %.4e" % r["value"], "median-based %.4e" % r["value_at_median"], {k: r["roofline"].get(k) for k in ("frac", "avg_launch_us", "min_us", "median_us", "max_us")})
for o in r["other_configs"]:
    if "error" in o: print("ERROR", o); continue
    print("%-60s %.4e (median %.4e) frac %.3f %s" % (o["workload"][:60], o["value"], o["value_at_median"], o["roofline"]["frac"], o["roofline"].get("program", "")))
print(r["cpu_baseline"])
PY
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
