#!/bin/bash
# GPU: C2 (jacobi2d 4096^2 f32) over fusion depth x chunk length x step order -- does a deeper group
# with longer chunks and independent stages (k1.rev=1) beat the depth-4 default?  (round 3)
# usage: bash tools/c2_sweep.sh > gpurun_out/c2_sweep.log
for fuse in 4 6 8; do
  for li in 0 23 32 48 64 96; do
    for rev in 0 1; do
      opts="fuse=$fuse;k1.rev=$rev"
      [ "$li" != "0" ] && opts="$opts;k2.li=$li"
      python bench.py --workload c2 --stages 240 --steps 3 --warmup 1 --no-cpu-baseline --options "$opts" 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln)
        print('$opts'.ljust(32), round(r['value']), 'Mcells/s', round(r['roofline']['avg_launch_us'], 2), 'us/launch', r['config']['schedule'][30:150])
"
    done
  done
done
