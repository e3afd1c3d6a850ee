#!/bin/bash
# Round 4, GPU session 37: float64 crosses with a second spatial field every other operator: compact groups two deep
# against one operator per launch.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab37
for o in "" "fuse=1" "fuse=2" "compact.prefer=0"; do
    timeout -k 10 200 python - "$o" <<'PY'
import sys, os, tempfile, re
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
from oracle import numpy_oracle as npo
opts = sys.argv[1]
for dt, shape in (("float64", "cross"), ("float32", "cross")):
    prog, _ = programs.synthesize(dt, 12, 0.5, 512, 512, 512, 1, 1, 1, stencil_shape=shape)
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
    pp = npo.load_program(prog)
    rng = np.random.default_rng(1)
    with Plan(lower(chain), options=opts) as plan:
        arrays = [rng.uniform(-1, 1, npo._dims_shape(pp, npo._input_dims(pp, n))).astype(npo._NP[pp["inputs"][n]["data_type"]]) for n in plan.input_names]
        plan.upload(arrays); plan.execute(1); plan.synchronize()
        plan.execute(3); plan.synchronize()
        ms = plan.elapsed_ms() / 3
        print("%-8s %-6s %-20s %9.0f Mcells/s  %s" % (dt, shape, opts, 12 * 134.217728 / ms * 1e3, plan.describe().split("\n")[1][9:150]), flush=True)
PY
done
