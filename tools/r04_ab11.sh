#!/bin/bash
# Round 4: fusion depth 3 for 3-D programs whose sums are typed float (integer boundary literals: the generator's
# programs) -- half the vector instructions per update of the double-typed jacobi.
OUT=gpurun_out/r04_ab11
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab11
for o in "" "fuse=3" "fuse=3;k1.skip=1" "fuse=4"; do
  for c in "cross 3-D f32" "diffusion 3-D f32" "hotspot 3-D f32" "fork 3-D f32"; do
    python tools/synth_perf.py --stages 12 --only "$c" --opts "$o" 2>/dev/null | grep Mcells | python3 -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln)
    if r['case'].startswith('$c') and ('extra' not in r['case']):
        print('%-22s %-20s launches %2d  %8d Mcells/s  %s' % (r['case'][:22], r['opts'], r['launches'], r['Mcells/s'], r['first'][7:100]))"
  done
done
