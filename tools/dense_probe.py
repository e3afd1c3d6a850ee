#!/usr/bin/env python3
"""GPU: the dense kernel's streaming forms on the generator's boxes (bin/synthesize.py of the reference, shape `box`) at
benchmark size -- time per launch for a list of plan-option variants, every result compared bit for bit with the same
program on the generic kernel (`generic_only=1`).
usage: dense_probe.py WORKLOAD [--variants "k1.bx=64;k1.by=2;k1.rj=4|dense.inslots=2|..."] [--stages N] [--reps N]
  WORKLOAD: box27 | box27_f64 | box9_2d | box125 | box125_f64 | box25_2d | box343 | box49_2d | cross2 | cross2_f64 | cross3
  a variant is a plan-option string; the empty variant is the planner's default.  SF_HIP_LIBNAME=libsf_hip_head.so in the
  environment runs the same variants on another build of the library (A/B on one box)."""
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

WORKLOADS = {
    "box27": ("float32", (512, 512, 512), 1, 4),
    "box27_f64": ("float64", (512, 512, 512), 1, 4),
    "box9_2d": ("float32", (4096, 4096, 0), 1, 4),
    "box125": ("float32", (512, 512, 512), 2, 2),
    "box125_f64": ("float64", (512, 512, 512), 2, 2),
    "box25_2d": ("float32", (4096, 4096, 0), 2, 4),
    "box343": ("float32", (512, 512, 512), 3, 2),
    "box49_2d": ("float32", (4096, 4096, 0), 3, 2),
    "cross2": ("float32", (512, 512, 512), 2, 4, "cross"),
    "cross2_f64": ("float64", (512, 512, 512), 2, 4, "cross"),
    "cross3": ("float32", (512, 512, 512), 3, 2, "cross"),
    "jacobi3d": ("float32", (512, 512, 512), 1, 6, "jacobi3d"),
    "diffusion1": ("float32", (512, 512, 512), 1, 6, "diffusion"),  # (a factor per term: scalars of the program)
    "diffusion2": ("float32", (512, 512, 512), 2, 4, "diffusion"),  # (the benchmark's operator: programs.jacobi3d)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(WORKLOADS))
    ap.add_argument("--variants", default="")
    ap.add_argument("--stages", type=int, default=0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--bc", default=None, help="boundary constant of every operator (default: the generator's 0)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--dims", default="", help="grid other than the workload's, e.g. 256,256,256")
    args = ap.parse_args()
    dtype, dims, extent, stages = WORKLOADS[args.workload][:4]
    stencil_shape = (WORKLOADS[args.workload] + ("box",))[4]
    if args.dims:
        dims = tuple(int(d) for d in args.dims.split(","))
    stages = args.stages or stages
    ext = [extent if d else 0 for d in dims]
    if stencil_shape == "jacobi3d":
        prog = programs.jacobi3d(tuple(dims), stages)
    else:
        prog, _ = programs.synthesize(dtype, stages, 0.0, *dims, *ext, stencil_shape=stencil_shape)
    if args.bc is not None:
        for k in prog["program"].values():
            for f in k["boundary_conditions"]:
                k["boundary_conditions"][f] = {"type": "constant", "value": json.loads(args.bc)}
    shape = [d for d in dims if d]
    x = np.random.default_rng(5).uniform(-1, 1, shape).astype(dtype)
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json"))))
    def set_scalars(plan):
        if plan.scalar_names:
            plan.set_scalars([float(str(prog["inputs"][n]["data"]).split(":")[-1]) for n in plan.scalar_names])

    want = None
    if not args.no_check:
        want = np.zeros(shape, x.dtype)
        with Plan(sfir, options="generic_only=1") as plan:
            set_scalars(plan)
            plan.run([x], [want], 1)
    for variant in args.variants.split("|"):
        try:
            plan = Plan(sfir, options=variant or None)
        except Exception as exc:  # noqa: BLE001
            print(json.dumps({"variant": variant, "error": str(exc)[:300]}), flush=True)
            continue
        got = np.zeros(shape, x.dtype)
        set_scalars(plan)
        plan.run([x], [got], 1)
        plan.upload([x])
        plan.execute(2)
        plan.synchronize()
        plan.set_profile(True)
        plan.execute(args.reps)
        plan.synchronize()
        times = plan.kernel_launch_times()
        plan.set_profile(False)
        # sustained: many executions back to back (the clock settles after a few milliseconds of load), wall clock
        plan.execute(args.reps)
        plan.synchronize()
        t0 = time.perf_counter()
        plan.execute(4 * args.reps)
        plan.synchronize()
        sustained_us = (time.perf_counter() - t0) / (4 * args.reps) / plan.num_launches * 1e6
        desc = plan.describe().splitlines()
        launches = plan.num_launches
        regs = {n: (r["vgprs"], r["lds"]) for n, r in plan.kernel_resources().items()}
        plan.close()
        line = {"variant": variant, "launches": launches}
        if want is not None:
            line["equal"] = bool(np.array_equal(got, want))
        for name, t in times.items():  # (min, median, max) in ms -> microseconds
            line[name] = [round(v * 1e3, 1) for v in t]
        line["sustained_us_per_launch"] = round(sustained_us, 1)
        line["vgprs, lds"] = regs
        line["launch"] = desc[1].strip()[desc[1].find("["):][:170]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
