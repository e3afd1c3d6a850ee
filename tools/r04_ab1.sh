#!/bin/bash
# Round 4, GPU session 1 (one box, one session):
#  (a) correctness of k1.skip=1 (halo rows no later stage reads are not evaluated): star / compact fuzz + the
#      full-size C3 parity test under SF_HIP_OPTIONS=k1.skip=1;
#  (b) A/B/C of the C3 launch on this box: the library of round 3's first profile (commit e047af9, kernel hash
#      44468cee), today's default, today's k1.skip=1 -- interleaved, three rounds;
#  (c) the 27-point box with and without k1.skip.
# usage (from the repository root, on the GPU box): bash tools/r04_ab1.sh
set -o pipefail
OUT=gpurun_out/r04_ab1
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab1
{
echo "== fuzz, k1.skip=1"
timeout -k 10 240 python tools/star_fuzz.py --seeds 400 --seconds 100 --options "k1.skip=1" 2>&1 | tail -4
timeout -k 10 240 python tools/star_fuzz.py --generator compact --seeds 200 --seconds 100 --options "k1.skip=1" 2>&1 | tail -4
} > $OUT/fuzz.log 2>&1 || { echo "fuzz failed"; tail -20 $OUT/fuzz.log; exit 1; }
cat $OUT/fuzz.log
echo "== full-size C3 + generator workloads under k1.skip=1"
SF_HIP_OPTIONS="k1.skip=1" timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "full_benchmark_configuration or full_size_generator_workloads or jacobi3d_chain_random or jacobi3d_tile_shapes" \
  > $OUT/pytest_skip.log 2>&1 || { echo "pytest failed"; tail -30 $OUT/pytest_skip.log; exit 1; }
tail -3 $OUT/pytest_skip.log
B="python bench.py --steps 10 --warmup 2 --no-other-configs --no-cpu-baseline"
for round in 1 2 3; do
  echo "== round $round"
  SF_HIP_LIBNAME=libsf_hip_e047af9.so $B > $OUT/c3_old_$round.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  $B --options "k1.skip=0" > $OUT/c3_new_$round.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  $B --options "k1.skip=1" > $OUT/c3_skip_$round.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  python - <<EOF
import json
for tag in ("old", "new", "skip"):
    r = json.load(open("$OUT/c3_%s_$round.json" % tag))
    print(tag, "%.4e Mcells/s" % r["value"], "avg launch %.2f us" % r["roofline"]["avg_launch_us"], r["roofline"]["kernel"])
EOF
done
echo "== box"
for round in 1 2; do
  for skip in 0 1; do
    python bench.py --workload box --stages 16 --steps 10 --warmup 2 --options "k1.skip=$skip" > $OUT/box_skip${skip}_$round.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
    python -c "
import json; r = json.load(open('$OUT/box_skip${skip}_$round.json'))
print('box skip=$skip', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['roofline']['kernel'])"
  done
done
