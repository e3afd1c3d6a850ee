#!/bin/bash
# Round 4, GPU session 27: where the fused dense launch (27-point box, 128x8 threads x 2 rows) spends its time
# (debug.whatif, timing only: 1 no barrier, 8 no loads, 16 no stores) and a few more shapes.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab27
export SF_HIP_SELF_CHECK=0
for round in 1 2; do
  for o in "debug.whatif=0" "debug.whatif=1" "debug.whatif=8" "debug.whatif=16" "debug.whatif=24" "debug.whatif=25"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "dense.t2=1;k1.bx=128;k1.by=8;k1.rj=2;$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-20s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'])"
  done
done
for o in "k1.bx=128;k1.by=8;k1.rj=2;k1.li=64" "k1.bx=128;k1.by=8;k1.rj=2;k1.li=32" "k1.bx=128;k1.by=8;k1.rj=2;k1.li=128" "k1.bx=128;k1.by=6;k1.rj=2" "k1.bx=128;k1.by=8;k1.rj=1" "k1.bx=128;k1.by=5;k1.rj=3"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "dense.t2=1;$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-40s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][30:130])"
done
