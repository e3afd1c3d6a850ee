#!/bin/bash
# Round 4, GPU session 30: fused dense form with ONE LDS slot for the input planes (a second barrier per step): taller
# tiles fit (18 rows: 32 row tiles x 16 chunks = 512 blocks on 256 units).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab30
SF_HIP_OPTIONS="dense.t2=2;dense.onein=1" timeout -k 10 100 python tools/star_fuzz.py --generator box_sum --seeds 400 --seconds 50 2>&1 | tail -1
for round in 1 2; do
  for o in "" "dense.onein=1" "k1.bx=128;k1.by=6;k1.rj=3" "k1.bx=128;k1.by=4;k1.rj=5" "k1.bx=128;k1.by=6;k1.rj=3;k1.li=32" "k1.bx=128;k1.by=10;k1.rj=2;allow_spills=1" "k1.bx=128;k1.by=9;k1.rj=2;allow_spills=1"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][7:130])"
  done
done
