#!/bin/bash
# Round 4, GPU session 12: compact3d.h issues the LDS reads of a source row one output row ahead (SF_LDS_AHEAD) and
# the first rows of a stage during the last row of the stage before.  Correctness (compact fuzz), then the 27-point
# box against HEAD's library on one box, interleaved.
set -o pipefail
OUT=gpurun_out/r04_ab12
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab12
timeout -k 10 200 python tools/star_fuzz.py --generator compact --seeds 300 --seconds 90 > $OUT/fuzz_compact.log 2>&1
echo "fuzz compact rc=$?"; tail -2 $OUT/fuzz_compact.log
for round in 1 2; do
  echo "== round $round"
  for lib in libsf_hip_head.so libsf_hip.so; do
    SF_HIP_LIBNAME=$lib python bench.py --workload box --stages 16 --steps 10 --warmup 2 > $OUT/box_${lib}_$round.json 2>$OUT/err.log || { echo "FAILED box $lib"; tail -5 $OUT/err.log; continue; }
    python -c "
import json; r = json.load(open('$OUT/box_${lib}_$round.json'))
print('box %-22s' % '$lib', '%.4e Mcells/s' % r['value'], 'avg launch %.2f us' % r['roofline']['avg_launch_us'], r['roofline']['kernel'])"
  done
done
