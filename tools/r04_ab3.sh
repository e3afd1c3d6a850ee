#!/bin/bash
# Round 4, GPU session 3: C3 launch time under k1.prio variants and selective non-temporal loads (k1.nt bit 2),
# interleaved, two rounds; fork program (16 operators of bin/synthesize.py -fork_frequency 0.25) with and without
# the depth-first operator order.
set -o pipefail
OUT=gpurun_out/r04_ab3
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab3
B="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  echo "== round $round"
  for opt in "k1.skip=0" "k1.prio=1" "k1.prio=2" "k1.prio=3" "k1.nt=5" "k1.nt=5;k1.prio=1" "k1.nt=4" "k1.nt=7"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c3_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -5 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c3_${tag}_$round.json'))
print('%-28s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['roofline']['kernel'])"
  done
done
timeout -k 10 200 python tools/star_fuzz.py --seeds 300 --seconds 60 --options "k1.nt=5;k1.prio=1" > $OUT/fuzz_nt5.log 2>&1
echo "fuzz rc=$?"; tail -2 $OUT/fuzz_nt5.log
python tools/synth_perf.py --fork > $OUT/fork.log 2>&1; echo "fork rc=$?"; tail -12 $OUT/fork.log
