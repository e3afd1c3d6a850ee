#!/bin/bash
# Round 4, GPU session 28: fused dense form with planes requested two steps ahead.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab28
timeout -k 10 300 python tools/dense_t2_check.py > gpurun_out/t2_check.log 2>&1; tail -1 gpurun_out/t2_check.log
for round in 1 2; do
  for o in "" "dense.t2=1" "dense.t2=1;k1.bx=128;k1.by=8;k1.rj=2;allow_spills=1" "dense.t2=1;k1.bx=128;k1.by=4;k1.rj=3" "dense.t2=1;k1.bx=128;k1.by=6;k1.rj=2" "dense.t2=1;k1.bx=128;k1.by=7;k1.rj=2"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-52s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][7:120])"
  done
done
for o in "" "dense.t2=1"; do
  timeout -k 10 120 python tools/synth_perf.py --only "box 2-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['operators'] == 16: print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], 'launches', r['launches'], r['first'][7:130])"
done
