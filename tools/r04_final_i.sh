#!/bin/bash
# Round 4, final GPU sequence, part I: a second fuzz campaign on seeds no earlier campaign of the round has seen
# (--first 2000), every generator, undivided and under slab decomposition; plan-time self-check on (default).
set -o pipefail
for g in star dag wide compact dense dense_sum box_sum copy; do
  timeout -k 10 140 python tools/star_fuzz.py --generator $g --first 2000 --seeds 400 --seconds 75 > gpurun_out/r04_final4_fuzz_$g.log 2>&1
  echo "fuzz $g rc=$? $(tail -1 gpurun_out/r04_final4_fuzz_$g.log)"
done
SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 140 python tools/star_fuzz.py --generator box_sum --first 2000 --seeds 400 --seconds 90 > gpurun_out/r04_final4_fuzz_box_sum_forced.log 2>&1
echo "fuzz box_sum (dense.t2=2) rc=$? $(tail -1 gpurun_out/r04_final4_fuzz_box_sum_forced.log)"
for g in mixed compact dense dag; do
  timeout -k 10 140 python tools/slab_fuzz.py --generator $g --first 2000 --seeds 200 --seconds 60 > gpurun_out/r04_final4_fuzz_slab_$g.log 2>&1
  echo "slab fuzz $g rc=$? $(tail -1 gpurun_out/r04_final4_fuzz_slab_$g.log)"
done
SF_HIP_OPTIONS="dense.t2=2" timeout -k 10 140 python tools/slab_fuzz.py --generator box_sum --first 2000 --seeds 200 --seconds 75 > gpurun_out/r04_final4_fuzz_slab_box_sum.log 2>&1
echo "slab fuzz box_sum rc=$? $(tail -1 gpurun_out/r04_final4_fuzz_slab_box_sum.log)"
