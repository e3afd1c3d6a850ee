#!/bin/bash
# GPU: one tools/dense_probe.py workload on two builds of the library (A/B on one box): libsf_hip_head.so (a copy of an
# earlier build, git-ignored) against libsf_hip.so.   usage: bash tools/ab_libs.sh <workload> <operators> ["variants"]
export SF_HIP_CACHE_DIR=off
wl=$1; ops=$2; variants=${3:-}
for lib in libsf_hip_head.so libsf_hip.so; do
  SF_HIP_LIBNAME=$lib timeout -k 10 300 python tools/dense_probe.py $wl --stages $ops --reps 20 --variants "$variants" 2>&1 | grep "^{" | OPS=$ops LIB=$lib WL=$wl python3 -c '
import sys, json, os
for l in sys.stdin:
    d = json.loads(l)
    print(os.environ["WL"], os.environ["LIB"], d["variant"], "equal", d.get("equal"), "launches", d["launches"], "us per operator %.1f" % (d["sustained_us_per_launch"] * d["launches"] / int(os.environ["OPS"])), d["launch"][:64])
'
done
