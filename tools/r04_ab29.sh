#!/bin/bash
# Round 4, GPU session 29: the dense kernel's fused form (dense.t2=1) on random chains of radius-1 plain sums
# (tests/random_programs.py: box_sum_program), undivided and under slab decomposition; the generator's boxes.
set -o pipefail
OUT=gpurun_out/r04_ab29
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab29
export SF_HIP_OPTIONS="dense.t2=2"
timeout -k 10 300 python tools/star_fuzz.py --generator box_sum --seeds 600 --seconds 180 > $OUT/fuzz_box_sum.log 2>&1
echo "fuzz box_sum rc=$? $(tail -1 $OUT/fuzz_box_sum.log)"
grep -c "dense" $OUT/fuzz_box_sum.log | head -1
timeout -k 10 300 python tools/slab_fuzz.py --generator box_sum --seeds 200 --seconds 150 > $OUT/slab_box_sum.log 2>&1
echo "slab fuzz box_sum rc=$? $(tail -2 $OUT/slab_box_sum.log | tr '\n' ' ')"
timeout -k 10 200 python tools/star_fuzz.py --generator compact --seeds 300 --seconds 60 > $OUT/fuzz_compact.log 2>&1
echo "fuzz compact rc=$? $(tail -1 $OUT/fuzz_compact.log)"
timeout -k 10 300 python tools/dense_t2_check.py > $OUT/t2_check.log 2>&1; tail -1 $OUT/t2_check.log
