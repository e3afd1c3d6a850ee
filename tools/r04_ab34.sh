#!/bin/bash
# Round 4, GPU session 34: the 27-point box in float64: compact kernel (three fused) against the dense kernel's fused form
# forced onto it (dense.t2=2: 256x4 threads x 2 rows, 8-row tiles).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab34
python - <<'PY'
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
prog, _ = programs.synthesize("float64", 12, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")
with tempfile.TemporaryDirectory() as tmp:
    chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
x = np.random.default_rng(1).uniform(-1, 1, prog["dimensions"])
for rnd in range(2):
    for opts in ({}, {"dense.t2": 2}, {"dense.t2": 2, "k1.bx": 256, "k1.by": 4, "k1.rj": 2}, {"dense.t2": 2, "k1.bx": 256, "k1.by": 3, "k1.rj": 3}, {"fuse": 2}):
        try:
            with Plan(lower(chain), options=opts) as plan:
                plan.upload([x]); plan.execute(1); plan.synchronize()
                plan.execute(3); plan.synchronize()
                ms = plan.elapsed_ms() / 3
                print("%-60s %.3f ms, %.3e Mcells/s  %s" % (opts, ms, 12 * 134.217728 / ms * 1e3, plan.describe().split("\n")[1][9:110]))
        except Exception as e:
            print(opts, "failed:", str(e)[:120])
PY
# (second part, after the planner took the results above: float64 boxes on the fused dense form 256x3x3, compact groups
#  two deep in float64) -- correctness on the new grouping
timeout -k 10 120 python tools/star_fuzz.py --generator compact --first 4000 --seeds 300 --seconds 70 2>&1 | tail -1
timeout -k 10 120 python tools/star_fuzz.py --generator box_sum --first 4000 --seeds 300 --seconds 60 2>&1 | tail -1
timeout -k 10 120 python tools/slab_fuzz.py --generator compact --first 4000 --seeds 100 --seconds 50 2>&1 | tail -1
timeout -k 10 300 python tools/dense_t2_check.py 2>&1 | tail -1
