#!/bin/bash
# Round 4, GPU session 8: the SLP vectoriser on / off for the star and wide-star kernels (it is off for compact since
# round 2 and for dense since this round), over the generator's workloads at benchmark size.
set -o pipefail
OUT=gpurun_out/r04_ab8
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab8
timeout -k 10 500 python tools/synth_perf.py > $OUT/synth_default.log 2>&1; echo "default rc=$?"
timeout -k 10 500 python tools/synth_perf.py --opts "star.slp=0;wide.slp=0" > $OUT/synth_noslp.log 2>&1; echo "noslp rc=$?"
python - <<'PY'
import json
def load(p):
    d = {}
    for ln in open(p):
        if ln.startswith("{"):
            r = json.loads(ln); d[r["case"]] = r
    return d
a, b = load("gpurun_out/r04_ab8/synth_default.log"), load("gpurun_out/r04_ab8/synth_noslp.log")
for k in a:
    if k in b:
        print("%-48s default %9d  no-slp %9d  (%+.1f %%)  launches %d" % (k[:48], a[k]["Mcells/s"], b[k]["Mcells/s"],
              100.0 * (b[k]["Mcells/s"] / a[k]["Mcells/s"] - 1), a[k]["launches"]))
PY
timeout -k 10 200 python tools/star_fuzz.py --seeds 300 --seconds 60 --options "star.slp=0" > $OUT/fuzz_noslp.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_noslp.log
timeout -k 10 200 python tools/star_fuzz.py --generator wide --seeds 300 --seconds 60 --options "wide.slp=0" > $OUT/fuzz_wide_noslp.log 2>&1; echo "fuzz wide rc=$?"; tail -1 $OUT/fuzz_wide_noslp.log
