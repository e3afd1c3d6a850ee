#!/bin/bash
# Round 4, GPU session 25: a star chain cut short by an operator only the compact kernel takes now joins it in one
# compact group (compact.prefer, default 1): the generator's chains with a second spatial field, with and without;
# star / compact / dag fuzz on the new grouping.
set -o pipefail
OUT=gpurun_out/r04_ab25
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab25
for g in star compact dag; do
  timeout -k 10 120 python tools/star_fuzz.py --generator $g --seeds 300 --seconds 60 > $OUT/fuzz_$g.log 2>&1
  echo "fuzz $g rc=$? $(tail -1 $OUT/fuzz_$g.log)"
done
for round in 1 2; do
  for o in "compact.prefer=0" "compact.prefer=1"; do
    timeout -k 10 200 python tools/synth_perf.py --only "extra field" --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('%-18s' % '$o', '%-44s' % r['case'][:44], '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], 'launches', r['launches'], r['first'][7:34])"
  done
done
