#!/bin/bash
# Round 4, final GPU sequence, part B: rocprofv3 evidence of the final code objects (tools/profile_round.sh), in two
# calls (gpurun's time limit): $1 = "1": c3 c2 c5 box; "2": wide dense fork generic + the slab kernels.
if [ "$1" = 1 ]; then
  SF_PROFILE_WORKLOADS="c3 c2 c5 box" SF_PROFILE_SLABS="none" bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_1.log 2>&1
else
  SF_PROFILE_WORKLOADS="wide dense fork generic" SF_PROFILE_SLABS="2 4 8" bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04_2.log 2>&1
fi
echo "profile_round part $1 rc=$?"
ls gpurun_out | grep "^r04_bench" | head -40
python3 -c "
import json; t = json.load(open('gpurun_out/hbm_traffic.json'))
for k, v in t.items(): print(k, v.get('round'), '%.4f GB' % (v['hbm_bytes_per_launch'] / 1e9), v.get('valu_busy'))"
