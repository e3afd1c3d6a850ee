#!/bin/bash
# Round 4, GPU session 20: C3 with 1024-thread blocks (four waves per SIMD at <= 128 registers: two rows per thread)
# against the default 128x4x5 (two waves per SIMD, 228 registers).
set -o pipefail
OUT=gpurun_out/r04_ab20
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab20
B="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  for opt in "" "k1.bx=128;k1.by=8;k1.rj=2" "k1.bx=128;k1.by=8;k1.rj=2;k1.pf2=0" "k1.bx=128;k1.by=8;k1.rj=2;k1.li=64" "k1.bx=128;k1.by=8;k1.rj=2;k1.li=128" "k1.bx=64;k1.by=16;k1.rj=2"; do
    tag=$(echo "x$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c3_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c3_${tag}_$round.json'))
print('%-44s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][:110])"
  done
done
