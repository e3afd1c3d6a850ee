#!/bin/bash
# Round 4, GPU session 33: C3 with 768-thread blocks (three waves per SIMD at <= 170 registers: 128x6 threads x 3 rows).
set -o pipefail
OUT=gpurun_out/r04_ab33
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab33
B="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  for opt in "" "k1.bx=128;k1.by=6;k1.rj=3" "k1.bx=128;k1.by=6;k1.rj=3;k1.pf2=0" "k1.bx=128;k1.by=6;k1.rj=3;k1.li=32" "k1.bx=128;k1.by=6;k1.rj=3;k1.li=64"; do
    tag=$(echo "x$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c3_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c3_${tag}_$round.json'))
print('%-44s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][:110])"
  done
done
