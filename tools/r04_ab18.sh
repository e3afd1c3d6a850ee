#!/bin/bash
# Round 4, GPU session 18: compact3d.h with the DPP moves of a stage step in one burst (k1.xbatch=1): compact fuzz, then
# the 27-point box against the default, interleaved.
set -o pipefail
OUT=gpurun_out/r04_ab18
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab18
timeout -k 10 200 python tools/star_fuzz.py --generator compact --seeds 300 --seconds 80 --options "k1.xbatch=1" > $OUT/fuzz_compact.log 2>&1
echo "fuzz compact rc=$?"; tail -1 $OUT/fuzz_compact.log
for round in 1 2; do
  for o in "k1.xbatch=0" "k1.xbatch=1" "k1.xbatch=1;k1.fence=0"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-30s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][:110])"
  done
done
for o in "k1.xbatch=0" "k1.xbatch=1"; do
timeout -k 10 120 python tools/synth_perf.py --only "box 2-D" --opts "$o" 2>/dev/null | grep "^{" | cut -c1-330
done
