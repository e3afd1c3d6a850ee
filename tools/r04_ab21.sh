#!/bin/bash
# Round 4, GPU session 21: the dense kernel's streaming form on sums over random SUBSETS of the 124 offsets ordered by
# plane (tests/random_programs.py: dense_sum_program since round 4): the parity tests, then a longer fuzz campaign.
set -o pipefail
OUT=gpurun_out/r04_ab21
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab21
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "plain_sums or dense or extent_two" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 300 python tools/star_fuzz.py --generator dense_sum --seeds 400 --seconds 200 > gpurun_out/r04_final2_fuzz_dense_sum.log 2>&1
echo "fuzz dense_sum rc=$? $(tail -1 gpurun_out/r04_final2_fuzz_dense_sum.log)"
