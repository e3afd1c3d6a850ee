#!/usr/bin/env python3
"""GPU: second look at the star-kernel code objects that spill SGPRs and give wrong
results (tools/spill_probe.py).  Only the shapes that failed there,
ONE fused launch (3 operators, fuse=3), and for every wrong plane what it holds
instead: zeros, another plane of the right answer, the answer with one input plane
missing ...  Compiler flags under test come in through SF_HIP_EXTRA_FLAGS.
(FAILING lists the shapes that failed with the kernel skeleton of mid round 2 -- git 54b8989 --, whose two LDS
reads under `if (ty ...)` were the trigger; with the skeleton that reads them unconditionally none of them fails,
profiles/r02_spill_probe_final_skeleton.log.  To see the fault again check that commit out.)
usage: SF_HIP_UNSAFE_SGPR_SPILLS=1 SF_HIP_REPORT_SGPR_SPILLS=1 spill_probe2.py [n0]"""
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

FAILING = [(64, 1, 7), (64, 2, 6), (64, 2, 8), (64, 4, 6), (64, 4, 8), (64, 8, 8), (128, 1, 8), (128, 2, 6),
           (128, 2, 8), (128, 4, 6), (128, 4, 8)]
CONTROL = [(64, 2, 7), (128, 2, 7), (64, 4, 5)]


def explain(got, want, x, prog, p):
    """What does wrong plane p hold?"""
    if not got[p].any():
        return "zeros"
    for q in range(want.shape[0]):
        if q != p and np.array_equal(got[p], want[q]):
            return "answer of plane %d" % q
    wrong = np.argwhere(got[p] != want[p])
    rows = sorted(set(int(w[0]) for w in wrong))
    cols = sorted(set(int(w[1]) for w in wrong))
    return "rows %s cols %d..%d (%d points)" % (rows if len(rows) < 12 else "%d..%d" % (rows[0], rows[-1]),
                                                 cols[0], cols[-1], len(wrong))


def main():
    n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    rng = np.random.default_rng(99)
    shape, stages = (n0, 30, 64), 3
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    print("# flags:", os.environ.get("SF_HIP_EXTRA_FLAGS", "(none)"), " n0:", n0, flush=True)
    failing = 0
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
        for bx, by, rj in FAILING + CONTROL:
            opt = {"fuse": 3, "k1.bx": bx, "k1.by": by, "k1.rj": rj, "allow_spills": 1}
            try:
                plan = Plan(sfir, options=opt)
            except (ValueError, RuntimeError) as exc:
                print("%dx%d rj %d: no plan (%s)" % (bx, by, rj, str(exc).splitlines()[0][:80]), flush=True)
                continue
            res = list(plan.kernel_resources().values())
            got = np.full_like(want, 7.0)
            plan.run([x], [got], 1)
            planes = [p for p in range(n0) if not np.array_equal(got[p], want[p])]
            failing += bool(planes)
            print("%-4s %3dx%d rj %d  launches %d  vgpr %d agpr %d vspill %d sgpr_spill %d  wrong planes %s" % (
                "FAIL" if planes else "ok", bx, by, rj, plan.num_launches, res[0]["vgprs"], res[0]["agprs"],
                res[0]["spills"], res[0]["scratch"], planes), flush=True)
            if planes and os.environ.get("SF_PROBE_SAVE"):
                os.makedirs(os.environ["SF_PROBE_SAVE"], exist_ok=True)
                np.savez(os.path.join(os.environ["SF_PROBE_SAVE"], "%dx%d_rj%d.npz" % (bx, by, rj)), x=x, got=got, want=want)
            for p in planes[:6]:
                print("      plane %d: %s" % (p, explain(got, want, x, prog, p)), flush=True)
            plan.close()
    print("# failing: %d of %d" % (failing, len(FAILING) + len(CONTROL)), flush=True)


if __name__ == "__main__":
    main()
