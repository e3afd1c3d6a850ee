#!/bin/bash
# Round 4, GPU session 9: C5 on the six-row tile the freed staging registers allow; f64 radius-2 stars two deep.
set -o pipefail
OUT=gpurun_out/r04_ab9
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab9
B5="python bench.py --workload c5 --stages 300 --steps 10 --warmup 2"
for round in 1 2; do
  for opt in "k1.pf2=0" "k1.bx=64;k1.by=8;k1.rj=6" "k1.bx=64;k1.by=8;k1.rj=5;k1.pf2=2" "k1.prio=1" "k1.nt=5"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    $B5 --options "$opt" > $OUT/c5_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c5_${tag}_$round.json'))
print('c5 %-40s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.1f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][30:120])"
  done
done
python tools/synth_perf.py --only "wide cross 3-D f64" > $OUT/w64_t1.log 2>&1; grep Mcells $OUT/w64_t1.log | cut -c1-220
python tools/synth_perf.py --only "wide cross 3-D f64" --opts "fuse=2" > $OUT/w64_t2.log 2>&1; grep Mcells $OUT/w64_t2.log | cut -c1-220
timeout -k 10 100 python tools/star_fuzz.py --generator wide --first 2000 --seeds 200 --seconds 60 --options "fuse=2" > $OUT/fuzz_wide_f2.log 2>&1; echo "fuzz wide fuse=2 rc=$?"; tail -1 $OUT/fuzz_wide_f2.log
