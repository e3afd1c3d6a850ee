#!/bin/bash
# Round 4, GPU session 36: 27-point boxes with a second spatial field every other operator (synthesize ... 0.5 ... box):
# compact groups two deep (12-row, 256-column tiles: 2.25 x redundant) against one operator per launch.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab36
for o in ""; do
  for dt in float32 float64; do
    timeout -k 10 200 python - "$o" $dt <<'PY'
import sys, os, tempfile, re
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
from oracle import numpy_oracle as npo
opts, dt = sys.argv[1], sys.argv[2]
prog, _ = programs.synthesize(dt, 12, 0.5, 512, 512, 512, 1, 1, 1, stencil_shape="box")
with tempfile.TemporaryDirectory() as tmp:
    chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
pp = npo.load_program(prog)
rng = np.random.default_rng(1)
try:
    with Plan(lower(chain), options=opts) as plan:
        arrays = [rng.uniform(-1, 1, npo._dims_shape(pp, npo._input_dims(pp, n))).astype(npo._NP[pp["inputs"][n]["data_type"]]) for n in plan.input_names]
        plan.upload(arrays); plan.execute(1); plan.synchronize()
        plan.execute(3); plan.synchronize()
        ms = plan.elapsed_ms() / 3
        kinds = {}
        for l in plan.describe().split("\n"):
            m = re.search(r"launch (sf_\w+?)_[0-9a-f]{8}", l)
            if m: kinds[m.group(1)] = kinds.get(m.group(1), 0) + 1
        print("%-8s %-46s %9.0f Mcells/s  %s" % (dt, opts, 12 * 134.217728 / ms * 1e3, kinds), flush=True)
except Exception as e:
    print(dt, opts, "FAILED", str(e)[:150])
PY
  done
done
