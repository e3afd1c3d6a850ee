#!/bin/bash
# Round 4, GPU session 19: streaming dense kernel, non-temporal loads / stores (k1.nt bit 0: stores, bit 1: loads).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab19
timeout -k 10 100 python tools/star_fuzz.py --generator dense_sum --seeds 300 --seconds 40 2>&1 | tail -1
for round in 1 2; do
  for o in "k1.nt=0" "k1.nt=1" "k1.nt=2" "k1.nt=3"; do
    timeout -k 10 120 python tools/synth_perf.py --only "big box 3-D" --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('%-24s' % '$o', '%8.0f Mcells/s' % r['Mcells/s'], 'ms/op %.3f' % (r['ms'] / r['operators']), r['first'][30:150])"
  done
done
