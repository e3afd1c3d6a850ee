#!/bin/bash
# Round 4, GPU session 17: streaming dense kernel, tile shapes x (next plane requested before / after the barrier).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab17
for round in 1 2; do
  for sh in "64;k1.by=2;k1.rj=4" "64;k1.by=4;k1.rj=2" "64;k1.by=2;k1.rj=2" "128;k1.by=2;k1.rj=4" "128;k1.by=1;k1.rj=4" "64;k1.by=4;k1.rj=4" "64;k1.by=2;k1.rj=3" "32;k1.by=4;k1.rj=4" "64;k1.by=1;k1.rj=4" "64;k1.by=2;k1.rj=6"; do
    for e in 0 1; do
      o="k1.bx=$sh;dense.early=$e"
      timeout -k 10 120 python tools/synth_perf.py --only "big box 3-D" --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('%-44s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms/op %.3f' % (r['ms'] / r['operators']), r['first'][30:150])"
    done
  done
done
