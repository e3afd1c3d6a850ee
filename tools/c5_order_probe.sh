#!/bin/bash
# C5 (and the wide / box workloads with k-tiles): logical tile order inside an XCD's share, k1.order=0|1.
# Time from bench.py, FETCH_SIZE from one PMC pass each.  usage (GPU box): bash tools/c5_order_probe.sh
export TMPDIR=/tmp SF_HIP_SELF_CHECK=0
torch_comgr=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))" 2>/dev/null)
[ -f "$torch_comgr" ] && export LD_PRELOAD="$torch_comgr${LD_PRELOAD:+:$LD_PRELOAD}"
out=gpurun_out/order_probe; rm -rf $out; mkdir -p $out
for wl in ${WORKLOADS:-c5}; do
for rep in 1 2; do
  for order in 0 1; do
    python3 bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --options "k1.order=$order" 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$wl order $order rep $rep: %.4f ms/step  %s' % (d['ms_per_step'], d['config'].get('schedule', '')[:110]))"
  done
done
for order in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_${wl}_$order -- python3 bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --options "k1.order=$order" > $out/pmc_${wl}_$order.log 2>&1
  python3 - <<PY
import csv, glob
vals = []
for p in glob.glob("$out/pmc_${wl}_$order/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("sf_"):
            vals.append(float(r["Counter_Value"]))
print("$wl order $order: FETCH_SIZE mean %.0f KB x2 = %.4f GB over %d dispatches" % (sum(vals) / max(1, len(vals)), 2 * 1024 * sum(vals) / max(1, len(vals)) / 1e9, len(vals)))
PY
done
done
