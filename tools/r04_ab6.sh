#!/bin/bash
# Round 4, GPU session 6: the DAG groups of kernels/star3d.h -- fuzz (against the oracle), the new tests, the fork
# programs' throughput with and without them.
set -o pipefail
OUT=gpurun_out/r04_ab6
mkdir -p $OUT
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab6
timeout -k 10 150 python tools/star_fuzz.py --seeds 400 --seconds 90 > $OUT/fuzz_star.log 2>&1; echo "fuzz star rc=$?"; tail -2 $OUT/fuzz_star.log
timeout -k 10 200 python tools/star_fuzz.py --generator dag --seeds 400 --seconds 120 > $OUT/fuzz_dag.log 2>&1; echo "fuzz dag rc=$?"; tail -3 $OUT/fuzz_dag.log
timeout -k 10 200 python tools/star_fuzz.py --generator dag --first 1000 --seeds 400 --seconds 120 --options "dag.windows=4" > $OUT/fuzz_dag_w4.log 2>&1; echo "fuzz dag w4 rc=$?"; tail -3 $OUT/fuzz_dag_w4.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "dag or fork_join or reference_test_programs or synthesized" > $OUT/pytest_dag.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_dag.log
python - > $OUT/fork_perf.log 2>&1 <<'PY'
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
rng = np.random.default_rng(5)
cases = [("fork 3-D f32", ("float32", 16, 0.0, 512, 512, 512, 1, 1, 1), ["dag=0", "", "dag.windows=3"]),
         ("fork 2-D f32", ("float32", 16, 0.0, 4096, 4096, 0, 1, 1, 0), ["dag=0", "", "dag.windows=4", "dag.windows=8"]),
         ("fork 3-D f64", ("float64", 16, 0.0, 512, 512, 512, 1, 1, 1), ["dag=0", "", "dag.windows=4"])]
with tempfile.TemporaryDirectory() as tmp:
    for label, pos, optlist in cases:
        prog, _ = programs.synthesize(*pos, fork_frequency=0.25)
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
        shape = prog["dimensions"]
        dtype = np.float32 if pos[0] == "float32" else np.float64
        x = rng.random(shape).astype(dtype)
        for opts in optlist:
            plan = Plan(lower(chain), options=opts or None)
            plan.upload([x])
            for _ in range(2):
                plan.execute(1); plan.synchronize()
            times = []
            for _ in range(5):
                plan.execute(4); plan.synchronize(); times.append(plan.elapsed_ms() / 4)
            ms = float(np.median(times))
            cells = float(np.prod(shape)) * len(prog["program"])
            print(json.dumps({"case": label, "opts": opts, "launches": plan.num_launches, "dag groups": plan.describe().count("[dag:"),
                              "ms": round(ms, 3), "Mcells/s": round(cells / ms / 1e3)}), flush=True)
            plan.close()
PY
echo "fork perf rc=$?"; cat $OUT/fork_perf.log | grep -v amdgpu.ids
