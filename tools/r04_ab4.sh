#!/bin/bash
# Round 4, GPU session 4: does skipping the unread halo rows pay once the waves that share a SIMD are balanced
# (k1.wmap), and does it make fusion depth 3 worth it on C3?  Box (VALU-heavy) and C3, interleaved twice.
set -o pipefail
OUT=gpurun_out/r04_ab4
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab4
timeout -k 10 200 python tools/star_fuzz.py --seeds 300 --seconds 70 --options "k1.skip=1;k1.wmap=1" > $OUT/fuzz_wmap.log 2>&1
echo "fuzz star rc=$?"; tail -2 $OUT/fuzz_wmap.log
timeout -k 10 200 python tools/star_fuzz.py --generator compact --seeds 300 --seconds 70 --options "k1.skip=1;k1.wmap=1" > $OUT/fuzz_wmap_compact.log 2>&1
echo "fuzz compact rc=$?"; tail -2 $OUT/fuzz_wmap_compact.log
B="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  echo "== round $round"
  for opt in "k1.skip=0" "k1.skip=1;k1.wmap=1" "fuse=3" "fuse=3;k1.skip=1;k1.wmap=1" "fuse=3;k1.skip=1" \
             "fuse=3;k1.pf2=0;k1.skip=1;k1.wmap=1;k1.bx=128;k1.by=4;k1.rj=5" "fuse=3;k1.pf2=0;k1.bx=128;k1.by=4;k1.rj=5"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c3_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -5 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c3_${tag}_$round.json'))
print('%-70s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][:90])"
  done
  for opt in "k1.skip=0" "k1.skip=1;k1.wmap=1" "k1.skip=1"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    python bench.py --workload box --stages 16 --steps 10 --warmup 2 --options "$opt" > $OUT/box_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED box $opt"; tail -5 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/box_${tag}_$round.json'))
print('box %-66s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'])"
  done
done
