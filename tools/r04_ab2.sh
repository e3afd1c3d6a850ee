#!/bin/bash
# Round 4, GPU session 2: (a) the two tests added for ADVICE r03 (zero-reach launch in the native schedule,
# bounded RCCL rung); (b) star fuzz with the L2 touch-ahead (k1.l2pf) on; (c) C3 launch time under
# k1.l2pf / k1.l2pfs / k1.prio, interleaved, two rounds.
set -o pipefail
OUT=gpurun_out/r04_ab2
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab2
timeout -k 10 400 python -m pytest tests/test_distributed.py -x -q -m gpu -k "zero_reach or bounds_a_stalled or rccl_rung_sends_to_itself" > $OUT/pytest_advice.log 2>&1
echo "pytest advice rc=$?"; tail -5 $OUT/pytest_advice.log
timeout -k 10 200 python tools/star_fuzz.py --seeds 300 --seconds 80 --options "k1.l2pf=2" > $OUT/fuzz_l2pf.log 2>&1
echo "fuzz rc=$?"; tail -3 $OUT/fuzz_l2pf.log
B="python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2; do
  echo "== round $round"
  for opt in "k1.skip=0" "k1.l2pf=1" "k1.l2pf=2" "k1.l2pf=3" "k1.l2pf=2;k1.l2pfs=128" "k1.l2pf=4;k1.l2pfs=128" "k1.l2pf=2;k1.l2pfs=256" "k1.prio=1" "k1.prio=1;k1.l2pf=2"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    $B --options "$opt" > $OUT/c3_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -5 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/c3_${tag}_$round.json'))
print('%-28s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['roofline']['kernel'])"
  done
done
