#!/bin/bash
# Round 4, GPU session 16: where does the streaming dense launch (125-point box) spend its time?  debug.whatif switches
# parts of the step off (wrong results, timing only): 1 no barrier, 2 no LDS writes, 8 no loads, 16 no stores.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab16
export SF_HIP_SELF_CHECK=0
for round in 1 2; do
  for w in 0 1 2 8 16 24 27; do
    timeout -k 10 120 python tools/synth_perf.py --only "big box 3-D" --opts "k1.bx=64;k1.by=2;k1.rj=4;debug.whatif=$w" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('whatif %2d' % $w, '%8.0f Mcells/s' % r['Mcells/s'], 'ms/op %.3f' % (r['ms'] / r['operators']))"
  done
done
