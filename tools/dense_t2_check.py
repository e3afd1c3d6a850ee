#!/usr/bin/env python3
"""GPU: the dense kernel's fused form (two radius-1 plain sums per launch, dense3d.h SF_DENSE_T2) on the generator's boxes
of extent 1 -- 27 points, 9 in 2-D -- over grids that fit a tile, need several, cut rows into k-tiles or leave most of a
tile empty, float32 / float64, 2-4 operators (an odd one out stays on the compact kernel), int and float boundary
literals; every result against the NumPy oracle.  dense.t2=2 forces the form onto grids it would not choose.
`cross2`: the generator's radius-2 crosses instead (round 5: reach two per operator, terms that join their plane late).
`cross1` / `box3`: radius-1 crosses / the boxes under dense.t2=3, fuse=3 -- three operators per launch.
usage (from the repository root, on a GPU box): python tools/dense_t2_check.py [cross2|cross1|box3]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
from oracle import numpy_oracle as npo
bad = 0
cases = []
MODE = sys.argv[1] if len(sys.argv) > 1 else "box"
CROSS2 = MODE == "cross2"
THREE = MODE in ("cross1", "box3")
for dims in [(20, 37, 72), (9, 14, 24), (33, 50, 512), (7, 16, 516), (12, 70, 1028), (70, 136), (40, 512), (25, 1032)]:
    for dtype in ("float32", "float64"):
        for stages in (2, 3, 4):
            for bc in (0, 0.5, -1):
                cases.append((dims, dtype, stages, bc))
rng = np.random.default_rng(5)
for dims, dtype, stages, bc in cases:
    if rng.random() < 0.5 and len(cases) > 60: continue
    full = list(dims) + [0] * (3 - len(dims))
    if CROSS2 and (len(dims) < 3 or dtype != "float32"):
        continue
    if THREE and (len(dims) < 3 or dtype != "float32"):
        continue
    ext = [(2 if CROSS2 else 1) if d else 0 for d in full]
    prog, _ = programs.synthesize(dtype, stages + (2 if THREE else 0), 0.0, *full, *ext, stencil_shape="cross" if (CROSS2 or MODE == "cross1") else "box")
    for k in prog["program"].values():
        for f in k["boundary_conditions"]:
            k["boundary_conditions"][f] = {"type": "constant", "value": bc}
    x = rng.uniform(-1, 1, dims).astype(dtype)
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))
    got = np.zeros(dims, dtype)
    with Plan(lower(chain), options={"dense.t2": 3, "fuse": 3} if THREE else {"dense.t2": 2}) as plan:
        d = plan.describe()
        used = "_t3_" in d if THREE else "_t2_" in d
        plan.run([x], [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    ok = np.array_equal(got, want, equal_nan=True)
    if not ok:
        bad += 1
        diff = np.argwhere(got != want)
        print("MISMATCH", dims, dtype, stages, bc, "t2" if used else "no-t2", len(diff), "points; first", diff[:3].tolist(), d.split("\n")[1][:160])
    else:
        print("ok", dims, dtype, stages, bc, "t2" if used else "no-t2")
print("failures", bad)
