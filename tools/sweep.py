#!/usr/bin/env python3
"""GPU tuning sweep: times the generated kernels for one program under several
option strings and checks every variant's output against the first one.

usage: sweep.py [--size 512] [--stages 8] [--kind jacobi3d|jacobi2d|c5] opt1 opt2 ...
       (an option string is "key=val;key=val"; "-" means defaults)"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--stages", type=int, default=8)
    ap.add_argument("--kind", default="jacobi3d")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--random", action="store_true")
    ap.add_argument("--shape", default="", help="explicit shape, e.g. 4096x512x512")
    ap.add_argument("--bc-int", action="store_true",
                    help="integer boundary literal (bin/synthesize.py convention): float32 sums")
    ap.add_argument("opts", nargs="*")
    args = ap.parse_args()
    n = args.size
    if args.shape:
        shape = tuple(int(v) for v in args.shape.split("x"))
        prog = (programs.jacobi3d if len(shape) == 3 else programs.jacobi2d)(shape, args.stages)
        dtype, bpu = np.float32, 8.0
    elif args.kind == "jacobi3d":
        shape = (n, n, n)
        prog = programs.jacobi3d(shape, args.stages, bc_value=0 if args.bc_int else 0.0)
        dtype, bpu = np.float32, 8.0
    elif args.kind == "jacobi2d":
        shape = (n, n)
        prog = programs.jacobi2d(shape, args.stages)
        dtype, bpu = np.float32, 8.0
    else:
        shape = (n, n, n)
        prog = programs.diffusion_advection_laplacian(shape, repeats=max(1, args.stages // 3))
        dtype, bpu = np.float64, 16.0
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        chain = sf.KernelChainGraph(path)
        sfir = lower(chain)
    nk = len(chain.kernel_nodes)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, shape).astype(dtype) if args.random else np.ones(shape, dtype)
    out = np.zeros(shape, dtype)
    base = None
    scal = [chain.inputs[s]["data"] for s in chain.inputs if not chain.inputs[s]["input_dims"]]
    for o in (args.opts or ["-"]):
        opt = None if o == "-" else o
        try:
            t0 = time.time()
            plan = Plan(sfir, options=opt)
            tc = time.time() - t0
            if scal:
                plan.set_scalars(scal)
            plan.upload([x])
            plan.execute(1)
            plan.synchronize()
            times = []
            for _ in range(args.reps):
                plan.execute(1)
                plan.synchronize()
                times.append(plan.elapsed_ms())
            plan.download([out])
            ms = float(np.median(times))
            cells = float(np.prod(shape)) * nk
            if base is None:
                base = out.copy()
                same = "base"
            else:
                same = "same" if np.array_equal(base, out) else "DIFF max|d|=%g" % np.abs(base - out).max()
            print(json.dumps({
                "opt": o, "ms": round(ms, 4), "min_ms": round(min(times), 4),
                "Mcells/s": round(cells / ms / 1e3, 1),
                "alg_GB/s": round(cells * bpu / ms / 1e6, 1),
                "frac_8TB": round(cells * bpu / ms / 1e6 / 8000, 4),
                "launches": plan.num_launches, "check": same,
                "compile_s": round(tc, 2),
                "sched": plan.describe().splitlines()[1].strip()[:160],
                "autotune": [l.strip()[:400] for l in plan.describe().splitlines() if l.strip().startswith("autotune")][:2]}),
                flush=True)
            plan.close()
        except Exception as exc:  # keep sweeping
            print(json.dumps({"opt": o, "error": str(exc)[:300]}), flush=True)


if __name__ == "__main__":
    main()
