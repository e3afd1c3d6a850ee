#!/usr/bin/env python3
"""GPU: the C3 and C5 launches timed in a process that never imports torch -- ROCm's own hipRTC and comgr compile
the kernels, as for a C embedder of the library -- beside what `python bench.py` (PyTorch's hipRTC + comgr) gets on the
same box (the two must agree within 2 %).  Prints one JSON line per case.
usage: SF_HIP_NO_TORCH=1 python tools/no_torch_bench.py          (the variable is set by the script if missing)"""
import json
import os
import sys

os.environ.setdefault("SF_HIP_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


def main():
    assert "torch" not in sys.modules, "this process must not import torch"
    import tempfile
    cases = [("c3", programs.jacobi3d((512, 512, 512), 100), np.float32, 50),
             ("c5", programs.diffusion_advection_laplacian((512, 512, 512), repeats=20), np.float64, 20)]
    rng = np.random.default_rng(1)
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog, dtype, launches in cases:
            chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, name + ".json")))
            plan = Plan(lower(chain))
            scalars = [chain.inputs[k]["data"] for k in plan.scalar_names]
            if scalars:
                plan.set_scalars(scalars)
            plan.upload([rng.random(prog["dimensions"]).astype(dtype)])
            for _ in range(3):
                plan.execute(1)
                plan.synchronize()
            times = []
            for _ in range(10):
                plan.execute(1)
                plan.synchronize()
                times.append(plan.elapsed_ms() / plan.num_launches * 1e3)
            print(json.dumps({"case": name, "torch_imported": "torch" in sys.modules, "launch_us_median": float(np.median(times)),
                              "launch_us_min": float(min(times)), "schedule": plan.describe().splitlines()[1].strip()[:120],
                              "compiler": plan.compiler()}), flush=True)
            plan.close()


if __name__ == "__main__":
    main()
