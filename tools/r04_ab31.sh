#!/bin/bash
# Round 4, GPU session 31: fused dense form, shapes small enough for TWO blocks per compute unit (<= 80 KB of LDS).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab31
for round in 1 2; do
  for o in "" "k1.bx=128;k1.by=5;k1.rj=2;dense.onein=1" "k1.bx=128;k1.by=4;k1.rj=2;dense.onein=1" "k1.bx=128;k1.by=3;k1.rj=3;dense.onein=1" "k1.bx=128;k1.by=2;k1.rj=5;dense.onein=1" "k1.bx=128;k1.by=5;k1.rj=2;dense.onein=0" "k1.bx=128;k1.by=6;k1.rj=3;k1.nt=0" "k1.bx=128;k1.by=6;k1.rj=3;k1.nt=3"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][7:130])"
  done
done
