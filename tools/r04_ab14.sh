#!/bin/bash
# Round 4, GPU session 14: the 27-point box without DPP (k1.xlane=1: ds_swizzle + row images) against DPP (k1.xlane=0),
# at the tile shapes the register budget allows; compact fuzz for correctness first.
set -o pipefail
OUT=gpurun_out/r04_ab14
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab14
timeout -k 10 200 python tools/star_fuzz.py --generator compact --seeds 300 --seconds 80 > $OUT/fuzz_compact.log 2>&1
echo "fuzz compact rc=$?"; tail -2 $OUT/fuzz_compact.log
for round in 1 2; do
  for o in "k1.xlane=0" "k1.xlane=1" "k1.xlane=1;k1.bx=128;k1.by=4;k1.rj=4;allow_spills=1" "k1.xlane=1;k1.bx=128;k1.by=4;k1.rj=5;allow_spills=1" \
           "k1.xlane=0;k1.bx=128;k1.by=4;k1.rj=3" "k1.xlane=0;k1.bx=128;k1.by=4;k1.rj=4"; do
    timeout -k 10 120 python tools/synth_perf.py --only "box 3-D f32" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if r['launches'] == 8: print('%-70s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][:110])"
  done
done
