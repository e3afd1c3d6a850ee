#!/usr/bin/env python3
"""Prototype of the time-skewed launch order (Infinity-Cache blocking), driven
from Python through sf_plan_execute_step_ranges.

A chain of L launches is normally run launch by launch over the whole field:
every launch streams the field from HBM and back.  Here the stream axis is cut
into slabs of W planes and launch l works on planes [s*W - l*D, (s+1)*W - l*D)
of slab s (D = deepest reach of a launch), slab by slab: launch l reads what
launch l-1 wrote moments ago, while it is still in the 256 MiB Infinity Cache.

usage: skew_probe.py [--shape 512x512x512] [--stages 200] [--w 64,96,128] [--opts "..."]
Prints one JSON line per variant; every variant is checked bit for bit against
the plain launch order."""
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


def skewed(plan, n0, reach, width):
    steps = plan.num_steps
    launches = 0
    nslabs = -(-(n0 + (steps - 1) * reach) // width)
    for s in range(nslabs):
        base = s * width
        for l in range(steps):
            lo = base - l * reach
            hi = lo + width
            if hi <= 0:
                break
            lo, hi = max(lo, 0), min(hi, n0)
            if lo < hi:
                plan.execute_step_ranges(l, lo, hi)
                launches += 1
    return launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512x512x512")
    ap.add_argument("--stages", type=int, default=200)
    ap.add_argument("--w", default="64,96,128")
    ap.add_argument("--opts", default="")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    shape = tuple(int(v) for v in args.shape.split("x"))
    prog = (programs.jacobi3d if len(shape) == 3 else programs.jacobi2d)(shape, args.stages)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        chain = sf.KernelChainGraph(path)
        sfir = lower(chain)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    plan = Plan(sfir, options=args.opts or None)
    reach = max(plan.step_halo(s)[1] for s in range(plan.num_steps))
    cells = float(np.prod(shape)) * args.stages
    base = np.zeros(shape, np.float32)
    out = np.zeros(shape, np.float32)

    def timed(fn):
        best = 1e30
        for _ in range(args.reps):
            plan.upload([x])
            t0 = time.perf_counter()
            extra = fn()
            plan.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, extra

    t, _ = timed(lambda: plan.execute(1))
    plan.download([base])
    print(json.dumps({"order": "plain", "ms": round(t * 1e3, 3), "Mcells/s": round(cells / t / 1e6, 1),
                      "launches": plan.num_steps, "sched": plan.describe().splitlines()[1].strip()[:150]}),
          flush=True)
    for w in [int(v) for v in args.w.split(",") if v]:
        t, launches = timed(lambda: skewed(plan, shape[0], reach, w))
        plan.download([out])
        same = "same" if np.array_equal(base, out) else "DIFF max|d|=%g" % np.abs(base - out).max()
        print(json.dumps({"order": "skewed", "W": w, "ms": round(t * 1e3, 3),
                          "Mcells/s": round(cells / t / 1e6, 1), "launches": launches, "check": same}),
              flush=True)
    plan.close()


if __name__ == "__main__":
    main()
