#!/usr/bin/env python3
"""GPU: run many pinned tile shapes of the star kernel on small random problems
and compare every result with the oracle (bit-exact for jacobi / the f64 chain,
1e-6 for the math program).  Prints one line per failing configuration."""
import itertools
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402
from oracle import numpy_oracle as npo  # noqa: E402

MATH = {
    "inputs": {"a": {"data": "constant:1.0", "data_type": "float32"}},
    "outputs": ["c"], "dimensions": [8, 16, 32],
    "program": {
        "b": {"computation_string": "b = sin(a[i,j,k]) * cos(a[i,j,k+1]) + sqrt(fabs(a[i-1,j,k]))",
              "boundary_conditions": {"a": {"type": "constant", "value": 0.5}}, "data_type": "float32"},
        "c": {"computation_string": "t = max(b[i,j,k], b[i,j-1,k]); c = t if t > 0.3 else min(t, 0.1) - exp(b[i,j,k])",
              "boundary_conditions": {"b": {"type": "constant", "value": 0.0}}, "data_type": "float32"}}}


def main():
    rng = np.random.default_rng(99)
    cases = []
    for shape, stages in [((14, 30, 64), 4), ((9, 21, 136), 3)]:
        prog = programs.jacobi3d(shape, stages, bc_value=0.25)
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
        cases.append(("jacobi%s" % (shape, ), prog, x, want, 0.0, None))
    x = rng.uniform(-1, 1, (8, 16, 32)).astype(np.float32)
    cases.append(("math", MATH, x, npo.run_reference(MATH, {"a": x})["c"], 1e-6, None))
    c5 = programs.diffusion_advection_laplacian((10, 18, 40))
    x5 = rng.uniform(-1, 1, (10, 18, 40))
    ins5 = {k: v["data"] for k, v in c5["inputs"].items() if k != "a"}
    want5 = npo.run_reference(c5, dict(ins5, a=x5))["lap"]
    cases.append(("c5", c5, x5, want5, 0.0, [ins5[k] for k in c5["inputs"] if k != "a"]))
    nfail = ntotal = nsgpr = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog, x, want, tol, scal in cases:
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            sfir = lower(sf.KernelChainGraph(path))
            quick = "--quick" in sys.argv
            space = itertools.product([1, 2, 3], [64, 128], [1, 2, 4] if quick else [1, 2, 3, 4, 8],
                                      [1, 2, 3, 5] if quick else [1, 2, 3, 4, 5, 6, 7, 8])
            for fuse, bx, by, rj in space:
                if bx * by > 1024 or by * rj - 2 * fuse < 1:
                    continue
                opt = {"fuse": fuse, "k1.bx": bx, "k1.by": by, "k1.rj": rj, "allow_spills": 1}
                try:
                    plan = Plan(sfir, options=opt)
                except ValueError:
                    continue
                res = list(plan.kernel_resources().values())
                if scal:
                    plan.set_scalars(scal)
                out = np.zeros_like(want)
                plan.run([x], [out], 1)
                plan.close()
                ntotal += 1
                # (SF_HIP_REPORT_SGPR_SPILLS=1: `scratch` carries the SGPR spill count; such objects run
                # since the criterion became "no allocator code ahead of an EXEC restore")
                if os.environ.get("SF_HIP_REPORT_SGPR_SPILLS"):
                    nsgpr += any(r["scratch"] % 1000 > 0 for r in res)
                if ntotal % 50 == 0:
                    print("# %d configurations run, %d failures so far" % (ntotal, nfail), flush=True)
                ok = np.array_equal(out, want) if tol == 0.0 else npo.arrays_match(want, out, tol)
                if not ok:
                    nfail += 1
                    print(json.dumps({"case": name, "opt": opt, "maxrel": npo.max_rel_err(want, out),
                                      "res": res}), flush=True)
    print("configs run: %d (%d of them with code objects that spill SGPRs), failures: %d" % (ntotal, nsgpr, nfail))


if __name__ == "__main__":
    main()
