#!/bin/bash
# Round 4, GPU session 7: the dense kernel's sum form -- accumulator pairs fenced term by term (dense.il), scalar adds
# (dense.slp=0) -- on the 125-point box 512^3 and the 25-point box 4096^2; correctness by the dense_sum fuzz.
set -o pipefail
OUT=gpurun_out/r04_ab7
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab7
for o in "dense.il=1" "dense.il=2" "dense.slp=0"; do
  timeout -k 10 120 python tools/star_fuzz.py --generator dense_sum --seeds 200 --seconds 45 --options "$o" > $OUT/fuzz.log 2>&1; echo "fuzz $o rc=$?"; tail -1 $OUT/fuzz.log
done
for round in 1 2; do
  echo "== round $round"
  for opt in "dense.il=0" "dense.il=1" "dense.il=2" "dense.il=3" "dense.il=5" "dense.slp=0" "dense.slp=0;dense.il=1" "dense.il=1;k1.bx=64;k1.by=8;k1.rj=2" "dense.il=2;k1.bx=32;k1.by=8;k1.rj=2"; do
    tag=$(echo "$opt" | tr ';=.' '___')
    python bench.py --workload dense --stages 4 --steps 10 --warmup 2 --options "$opt" > $OUT/dense_${tag}_$round.json 2>$OUT/err.log || { echo "FAILED $opt"; tail -3 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/dense_${tag}_$round.json'))
print('%-44s' % '$opt', '%.4e Mcells/s' % r['value'], 'avg launch %.1f us' % r['roofline']['avg_launch_us'], r['config']['schedule'][30:110])"
  done
done
python tools/synth_perf.py --only "big box 2-D" > $OUT/box2d.log 2>&1; grep Mcells $OUT/box2d.log | cut -c1-200
python tools/synth_perf.py --only "big box 2-D" --opts "dense.il=1" > $OUT/box2d_il1.log 2>&1; grep Mcells $OUT/box2d_il1.log | cut -c1-200
python tools/synth_perf.py --only "big box 2-D" --opts "dense.il=2" > $OUT/box2d_il2.log 2>&1; grep Mcells $OUT/box2d_il2.log | cut -c1-200
