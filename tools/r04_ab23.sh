#!/bin/bash
# Round 4, GPU session 23: C2 (jacobi2d 4096^2 f32 x 1000): chunk length and fusion depth around the planner's choice.
set -o pipefail
OUT=gpurun_out/r04_ab23
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab23
B="python bench.py --workload c2 --steps 10 --warmup 3 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  for opt in "" "k2.li=16" "k2.li=32" "k2.li=46" "k2.li=64" "fuse=3" "fuse=5" "fuse=6" "fuse=8;k2.li=46" "fuse=6;k2.li=32" "k1.vk=2" "k2.bx=128"; do
    tag=$(echo "x$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c2_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c2_${tag}_$round.json'))
print('%-22s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][:120])"
  done
done
