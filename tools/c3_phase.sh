#!/bin/bash
# GPU: VERDICT r04 next 7 -- is the C3 launch held back by its eight plane sweeps (one chunk of 64 planes per XCD) running
# exactly 64 MiB apart, i.e. in the same phase of the HBM channel interleave?  If so, chunks of 65 / 66 / 68 / 72 planes
# (sweeps 65 .. 72 MiB apart; 1.5-12 % more planes on the critical path) would beat 64 despite the longer path.
# usage: bash tools/c3_phase.sh <tag>
tag=${1:-r05}
for li in 64 65 66 68 72 64; do
  python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --options "k1.li=$li" 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.readline())
print('k1.li=$li', 'Mcells/s %.4e' % r['value'], 'avg launch us %.2f' % r['roofline']['avg_launch_us'], 'min/median/max', [round(r['roofline'].get(k, 0), 1) for k in ('min_us', 'median_us', 'max_us')], r['config'].get('schedule', '')[:80])
"
done > gpurun_out/${tag}_c3_phase.log 2>&1
cat gpurun_out/${tag}_c3_phase.log
