#!/bin/bash
# Round 4, GPU session 24: star kernels on float-typed programs (the generator's: f32 adds) with the lane exchange through
# ds_bpermute (k1.dpp=0: no DPP in the loop, so two waves' f32 instructions may overlap) against DPP (default).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab24
for round in 1 2; do
  for o in "" "k1.dpp=0"; do
    timeout -k 10 300 python tools/synth_perf.py --stages 16 --opts "$o" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        if 'star' in r['first'].split('[')[1][:12] and 'wide' not in r['first']:
            print('%-10s' % '$o', '%-44s' % r['case'][:44], '%8.0f Mcells/s' % r['Mcells/s'], 'ms %.3f' % r['ms'], r['first'][7:34])"
  done
done
