#!/bin/bash
# Round 4, GPU session 26: the 27-point box (and the 9-point 2-D box) on the dense kernel's fused streaming form
# (dense.t2=1: two operators per launch, no lane exchange) against the compact kernel; tile shapes.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab26
for round in 1 2; do
  for o in "" "dense.t2=1" "dense.t2=1;k1.bx=64;k1.by=4;k1.rj=4" "dense.t2=1;k1.bx=64;k1.by=8;k1.rj=2" "dense.t2=1;k1.bx=128;k1.by=2;k1.rj=4" "dense.t2=1;k1.bx=128;k1.by=4;k1.rj=2" "dense.t2=1;k1.bx=128;k1.by=4;k1.rj=3" "dense.t2=1;k1.bx=64;k1.by=2;k1.rj=4" "dense.t2=1;k1.bx=128;k1.by=8;k1.rj=2"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][7:130])"
  done
done
for o in "" "dense.t2=1" "dense.t2=1;k2.bx=128" "dense.t2=1;k2.bx=256"; do
  timeout -k 10 120 python tools/synth_perf.py --only "box 2-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['operators'] == 16: print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], 'launches', r['launches'], r['first'][7:130])"
done
