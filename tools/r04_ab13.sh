#!/bin/bash
# Round 4, GPU session 13: where does the 27-point box launch wait?  Timing only (debug.whatif switches parts of the step
# off; the results of those runs are wrong by construction): 1 no barrier, 2 no LDS reads, 4 no lane exchange (DPP),
# 8 no loads of the streamed planes, 16 no stores, 32 nothing published.
set -o pipefail
OUT=gpurun_out/r04_ab13
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab13
export SF_HIP_SELF_CHECK=0   # the what-if kernels are wrong by construction
for round in 1 2; do
  for w in 0 16 24 8 59 63 43; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "debug.whatif=$w" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); print('whatif %2d' % $w, '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], 'launches', r['launches'])"
  done
done
