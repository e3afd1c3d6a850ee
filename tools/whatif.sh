#!/bin/bash
# GPU: timing-only code objects of a dense kernel (tools/whatif_objects.py builds them on the CPU under results/whatif/<dir>):
# sustained time per launch of each variant, and -- with `clock` -- the clock the chip holds while it runs
# (GRBM_GUI_ACTIVE / 8 / duration).  The results of these objects are WRONG by construction; nothing of them is reachable
# through sf_plan_create.
# usage: bash tools/whatif.sh <dir under results/whatif> <workload of tools/dense_probe.py> "<plan options>" [clock]
export SF_HIP_CACHE_DIR=off SF_HIP_SELF_CHECK=0 TMPDIR=/tmp
dir=$1; wl=$2; opts=${3:-}; mode=${4:-time}
for v in asis carry ahead2 ahead3 nobar nolds nodma nostore nomem valu; do
  [ -d results/whatif/$dir/$v ] || continue
  if [ "$mode" = clock ]; then
    out=gpurun_out/whatif_clock_${dir}_$v
    rm -rf $out
    SF_HIP_OBJECT_DIR=results/whatif/$dir/$v timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out -- python3 tools/dense_probe.py $wl --stages 8 --reps 8 --no-check --variants "$opts" > $out.log 2>&1
    python3 - $out $dir $v <<'PY'
import csv, glob, sys, statistics
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "dense" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
clk = []
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dense" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            clk.append((float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]] / 1e3, dur[r["Dispatch_Id"]]))
clk = clk[len(clk) // 2:]  # the later launches: the clock has settled
print(sys.argv[2], sys.argv[3], "launches", len(clk), "median GHz %.2f" % statistics.median(c for c, _ in clk), "median us %.1f" % statistics.median(t for _, t in clk))
PY
  else
    echo "== $dir $v"
    SF_HIP_OBJECT_DIR=results/whatif/$dir/$v timeout -k 10 120 python tools/dense_probe.py $wl --stages 8 --no-check --variants "$opts" 2>&1 | grep -v "^W\|amdgpu.ids" | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line[:200]); continue
    d=json.loads(line); ks=[k for k in d if k.startswith('sf_')]
    print('sustained us/launch', d.get('sustained_us_per_launch'), 'min/median/max', [d[k] for k in ks], d['launch'][:60])
"
  fi
done
