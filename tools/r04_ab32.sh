#!/bin/bash
# Round 4, GPU session 32: radius-3 boxes (the generator's extent 3) on the dense kernel's streaming form: parity tests,
# then 512^3 / 4096^2 against the generic kernel (dense.r3=0).
set -o pipefail
export SF_HIP_CACHE_DIR=$PWD/gpurun_out/cache_ab32
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "extent_three or extent_two" > gpurun_out/r04_ab32_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_ab32_pytest.log
python - <<'PY'
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
for label, args in (("343-point box 512^3 f32", ("float32", 2, 0.0, 512, 512, 512, 3, 3, 3)), ("49-point box 4096^2 f32", ("float32", 4, 0.0, 4096, 4096, 0, 3, 3, 0))):
    prog, _ = programs.synthesize(*args, stencil_shape="box")
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
    dims = prog["dimensions"]
    x = np.random.default_rng(1).uniform(-1, 1, dims).astype(np.float32)
    for opts in ({}, {"dense.r3": 0}):
        with Plan(lower(chain), options=opts) as plan:
            plan.upload([x])
            plan.execute(1)
            plan.synchronize()
            reps = 3
            plan.execute(reps)
            plan.synchronize()
            ms = plan.elapsed_ms() / reps / len(prog["program"])
            print("%-26s %-16s %.3f ms per operator, %.3e Mcells/s  %s" % (label, opts, ms, np.prod(dims) / ms / 1e3, plan.describe().split("\n")[1][9:120]))
PY
