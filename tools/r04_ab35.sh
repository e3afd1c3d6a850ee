#!/bin/bash
# Round 4, GPU session 35: a sweep over generator configurations nobody had timed (float64 variants, 2-D variants, extra
# fields): looking for plans whose throughput is out of line with their siblings'.
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab35
python - <<'PY'
import sys, os, tempfile, re
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
n, st = 512, 12
cases = []
for dt in ("float32", "float64"):
    for shape in ("cross", "box", "diffusion", "hotspot"):
        cases.append(("%s 3-D %s" % (shape, dt), (dt, st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": shape}))
        cases.append(("%s 2-D %s" % (shape, dt), (dt, st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0), {"stencil_shape": shape}))
    cases.append(("cross 3-D %s + extra" % dt, (dt, st, 0.5, n, n, n, 1, 1, 1), {}))
    cases.append(("box 3-D %s + extra" % dt, (dt, st, 0.5, n, n, n, 1, 1, 1), {"stencil_shape": "box"}))
    cases.append(("wide cross 3-D %s" % dt, (dt, st, 0.0, n, n, n, 2, 2, 2), {}))
    cases.append(("wide diffusion 3-D %s" % dt, (dt, st, 0.0, n, n, n, 2, 2, 2), {"stencil_shape": "diffusion"}))
    cases.append(("big box 3-D %s" % dt, (dt, 4, 0.0, n, n, n, 2, 2, 2), {"stencil_shape": "box"}))
    cases.append(("big box 2-D %s" % dt, (dt, 8, 0.0, 8 * n, 8 * n, 0, 2, 2, 0), {"stencil_shape": "box"}))
for label, args, kw in cases:
    try:
        prog, _ = programs.synthesize(*args, **kw)
        with tempfile.TemporaryDirectory() as tmp:
            chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
        p = prog
        ins = []
        rng = np.random.default_rng(1)
        with Plan(lower(chain)) as plan:
            from oracle import numpy_oracle as npo
            pp = npo.load_program(prog)
            arrays = []
            for name in plan.input_names:
                dims = npo._input_dims(pp, name)
                arrays.append(rng.uniform(-1, 1, npo._dims_shape(pp, dims)).astype(npo._NP[pp["inputs"][name]["data_type"]]))
            if plan.scalar_names:
                plan.set_scalars([pp["inputs"][nm]["data"] for nm in plan.scalar_names])
            plan.upload(arrays); plan.execute(1); plan.synchronize()
            plan.execute(3); plan.synchronize()
            ms = plan.elapsed_ms() / 3
            cells = float(np.prod(prog["dimensions"]))
            kinds = {}
            for l in plan.describe().split("\n"):
                m = re.search(r"launch (sf_\w+?)_[0-9a-f]{8}", l)
                if m: kinds[m.group(1)] = kinds.get(m.group(1), 0) + 1
            print("%-30s ops %2d  %9.0f Mcells/s  %s" % (label, len(prog["program"]), len(prog["program"]) * cells / ms / 1e3, kinds), flush=True)
    except Exception as e:
        print("%-30s FAILED %s" % (label, str(e)[:140]), flush=True)
PY
