#!/bin/bash
# Round 4, GPU session: C2 with windows of converted values (k2.wide) -- correctness (2-D star fuzz) and launch time.
set -o pipefail
OUT=gpurun_out/r04_ab10
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab10
timeout -k 10 200 python tools/star_fuzz.py --first 9000 --seeds 600 --seconds 90 --options "k2.wide=1" > $OUT/fuzz_wide_windows.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_wide_windows.log
B="python bench.py --workload c2 --steps 10 --warmup 2"
for round in 1 2; do
  for opt in "k2.wide=0" "k2.wide=1" "k2.wide=1;k1.pfd=1" "k1.pf2=1;k1.pfd=3"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c2_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c2_${tag}_$round.json'))
print('c2 %-28s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][30:110])"
  done
done
python tools/synth_perf.py --only "2-D f32" > $OUT/synth2d.log 2>&1; grep Mcells $OUT/synth2d.log | cut -c1-140
python tools/synth_perf.py --only "2-D f32" --opts "k2.wide=1" > $OUT/synth2d_wide.log 2>&1; grep Mcells $OUT/synth2d_wide.log | cut -c1-140
