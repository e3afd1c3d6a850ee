#!/usr/bin/env python3
"""CPU: which code objects does the planner pick for the measured workloads, and what do they
report (registers, SGPR spills, EXEC restores behind allocator code)?  Compile only, no GPU.
usage: [SF_HIP_STRICT_SGPR_SPILLS=1] selection_report.py [--size 512] [--only text]"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SF_HIP_REPORT_SGPR_SPILLS"] = "1"
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    n, st = args.size, 16
    syn = programs.synthesize
    cases = [
        ("C3 jacobi3d f32", programs.jacobi3d((n, n, n), 16), None),
        ("C2 jacobi2d f32", programs.jacobi2d((8 * n, 8 * n), 16), None),
        ("C5 chain f64", programs.diffusion_advection_laplacian((n, n, n), repeats=2), None),
        ("generic jacobi3d", programs.jacobi3d((n, n, n), 4), "generic_only=1"),
        ("cross 3-D f64", syn("float64", st, 0.0, n, n, n, 1, 1, 1)[0], None),
        ("diffusion 3-D f32", syn("float32", st, 0.0, n, n, n, 1, 1, 1, stencil_shape="diffusion")[0], None),
        ("hotspot 3-D f32", syn("float32", st, 0.0, n, n, n, 1, 1, 1, stencil_shape="hotspot")[0], None),
        ("box 3-D f32", syn("float32", st, 0.0, n, n, n, 1, 1, 1, stencil_shape="box")[0], None),
        ("cross 3-D + extra field", syn("float32", st, 0.5, n, n, n, 1, 1, 1)[0], None),
        ("cross 2-D f32", syn("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0)[0], None),
        ("hotspot 2-D f32", syn("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0, stencil_shape="hotspot")[0], None),
        ("box 2-D f32", syn("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0, stencil_shape="box")[0], None),
        ("cross 2-D + extra field", syn("float32", st, 0.5, 8 * n, 8 * n, 0, 1, 1, 0)[0], None),
    ]
    with tempfile.TemporaryDirectory() as tmp:
        for label, prog, opts in cases:
            if args.only and args.only not in label:
                continue
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            plan = Plan(lower(sf.KernelChainGraph(path)), options=opts)
            res = plan.kernel_resources()
            launched = [ln.strip() for ln in plan.describe().splitlines() if "sf_" in ln]
            print("== %s: %d launches" % (label, plan.num_launches))
            for name, r in res.items():
                used = any(name in ln for ln in launched)
                print("   %s %-34s vgpr %3d agpr %3d vspill %3d sgpr spills %2d late exec restores %d" % (
                    "*" if used else " ", name, r["vgprs"], r["agprs"], r["spills"], r["scratch"] % 1000,
                    r["scratch"] // 1000))
            plan.close()


if __name__ == "__main__":
    main()
