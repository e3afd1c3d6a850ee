#!/usr/bin/env python3
"""GPU: pinned tile shapes of the COMPACT kernel (kernels/compact3d.h) on small problems, every
result compared with the oracle bit for bit -- the counterpart of tools/config_fuzz.py, which pins
shapes of the star kernel.  Programs: the reference generator's 27-point box, a cross chain with an
extra streamed field every second operator, and two random compact chains (tests/random_programs.py).
Shapes that need more registers than the chip has spill; objects that show the compiler fault of
DESIGN.md 5.1 are refused by the library (the group is shortened) -- wrong results must not occur.
usage: compact_config_fuzz.py [--quick]"""
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
from tests.random_programs import compact_program  # noqa: E402


def case_inputs(prog, rng):
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        idims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, idims)).astype(npo._NP[desc["data_type"]])
                     if idims else desc["data"])
    return ins


def main():
    quick = "--quick" in sys.argv
    rng = np.random.default_rng(7)
    progs = [("box", programs.synthesize("float32", 4, 0.0, 12, 22, 72, 1, 1, 1, stencil_shape="box")[0]),
             ("cross + extra field", programs.synthesize("float32", 4, 0.5, 10, 26, 64, 1, 1, 1)[0])]
    for seed in (6, 18, 8):
        prog = compact_program(seed)
        if len(prog["dimensions"]) == 3:
            progs.append(("random compact chain %d" % seed, prog))
    nfail = ntotal = nflagged = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog in progs:
            ins = case_inputs(prog, rng)
            want = npo.run_reference(prog, inputs=ins)
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            chain = sf.KernelChainGraph(path)
            sfir = lower(chain)
            space = itertools.product([1, 2, 3], [64, 128], [1, 2, 4] if quick else [1, 2, 3, 4, 8],
                                      [1, 3, 5] if quick else [1, 2, 3, 4, 5, 6, 7])
            for fuse, bx, by, rj in space:
                if bx * by > 1024 or by * rj - 2 * fuse < 1:
                    continue
                opt = {"fuse": fuse, "k1.bx": bx, "k1.by": by, "k1.rj": rj, "allow_spills": 1}
                try:
                    plan = Plan(sfir, options=opt)
                except ValueError:
                    continue
                res = list(plan.kernel_resources().values())
                arrays = [np.ascontiguousarray(ins[n]) for n in plan.input_names]
                if plan.scalar_names:
                    plan.set_scalars([float(ins[n]) for n in plan.scalar_names])
                outs = [np.zeros_like(want[n]) for n in plan.output_names]
                plan.run(arrays, outs, 1)
                compact = "compact" in plan.describe()
                plan.close()
                ntotal += 1
                # (SF_HIP_REPORT_SGPR_SPILLS=1: `scratch` = SGPR spills + 1000 x flagged EXEC restores)
                nflagged += any(r["scratch"] >= 1000 for r in res) if os.environ.get("SF_HIP_REPORT_SGPR_SPILLS") else 0
                if ntotal % 50 == 0:
                    print("# %d configurations run, %d failures so far" % (ntotal, nfail), flush=True)
                ok = all(np.array_equal(o, want[n]) for o, n in zip(outs, plan.output_names))
                if not ok:
                    nfail += 1
                    print(json.dumps({"case": name, "opt": opt, "compact": compact, "res": res}), flush=True)
    print("configs run: %d (%d with a compiled object the library refused), failures: %d" % (ntotal, nflagged, nfail))


if __name__ == "__main__":
    main()
