#!/bin/bash
# Round 4, GPU session 15: dense3d.h's streaming form (every plane read from LDS once, five open output planes):
# correctness (dense_sum and dense fuzz), then the 125-point box against the form of round 3/4 (dense.stream=0).
set -o pipefail
OUT=gpurun_out/r04_ab15
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab15
timeout -k 10 200 python tools/star_fuzz.py --generator dense_sum --seeds 300 --seconds 90 > $OUT/fuzz_dense_sum.log 2>&1
echo "fuzz dense_sum rc=$?"; tail -2 $OUT/fuzz_dense_sum.log
timeout -k 10 200 python tools/star_fuzz.py --generator dense --seeds 300 --seconds 60 > $OUT/fuzz_dense.log 2>&1
echo "fuzz dense rc=$?"; tail -2 $OUT/fuzz_dense.log
for round in 1 2; do
  for o in "dense.stream=0" "" "k1.bx=64;k1.by=4;k1.rj=2" "k1.bx=64;k1.by=8;k1.rj=2" "k1.bx=64;k1.by=2;k1.rj=4" "k1.bx=32;k1.by=8;k1.rj=2" "k1.bx=64;k1.by=4;k1.rj=3"; do
    timeout -k 10 120 python tools/synth_perf.py --only "big box 3-D" --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('%-40s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms/op %.3f' % (r['ms'] / r['operators']), r['first'][:120])"
  done
done
timeout -k 10 120 python tools/synth_perf.py --only "big box 2-D" 2>/dev/null | grep "^{" | cut -c1-300
timeout -k 10 120 python tools/synth_perf.py --only "big box 2-D" --opts "dense.stream=0" 2>/dev/null | grep "^{" | cut -c1-300
