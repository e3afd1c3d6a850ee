#!/usr/bin/env python3
"""GPU: throughput of programs from the workload generator (the reference's
bin/synthesize.py conventions: integer boundary literals -> float32 sums,
coefficient 1/n, optional extra fields) at benchmark size.
usage: synth_perf.py [--size 512] [--stages 16] [--only hotspot] [--opts "fuse=3"]"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--stages", type=int, default=16)
    ap.add_argument("--only", default="", help="run the cases whose label contains this text")
    ap.add_argument("--opts", default="", help="plan options, e.g. fuse=3;k1.li=32")
    ap.add_argument("--fork", action="store_true",
                    help="only the fork / join programs (-fork_frequency 0.25), each with reorder=0 and reorder=1")
    args = ap.parse_args()
    n, st = args.size, args.stages
    cases = [
        ("cross 3-D f32", ("float32", st, 0.0, n, n, n, 1, 1, 1), {}),
        ("cross 3-D f64", ("float64", st, 0.0, n, n, n, 1, 1, 1), {}),
        ("diffusion 3-D f32", ("float32", st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": "diffusion"}),
        ("hotspot 3-D f32", ("float32", st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": "hotspot"}),
        ("box 3-D f32", ("float32", st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": "box"}),
        ("cross 3-D f32, extra field every 2nd stage", ("float32", st, 0.5, n, n, n, 1, 1, 1), {}),
        # 2-D programs are given as size_x, size_y, 0 (the reference generator pairs sizes
        # and extents by position: `0 N N` with extents `0 1 1` -- round 1's 2-D cases --
        # yields a k-only stencil on a 2-D grid)
        ("cross 2-D f32", ("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0), {}),
        ("hotspot 2-D f32", ("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0), {"stencil_shape": "hotspot"}),
        ("box 2-D f32", ("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0), {"stencil_shape": "box"}),
        ("cross 2-D f32, extra field every 2nd stage", ("float32", st, 0.5, 8 * n, 8 * n, 0, 1, 1, 0), {}),
        ("k-only 2-D f32 (round 1's 'cross 2-D')", ("float32", st, 0.0, 0, 8 * n, 8 * n, 0, 1, 1), {}),
        # radius-2 stars (extent 2 of the reference generator): kernels/wstar3d.h since round 3
        ("wide cross 3-D f32 (radius 2)", ("float32", st, 0.0, n, n, n, 2, 2, 2), {}),
        ("wide diffusion 3-D f32 (radius 2)", ("float32", st, 0.0, n, n, n, 2, 2, 2), {"stencil_shape": "diffusion"}),
        ("wide cross 3-D f64 (radius 2)", ("float64", st, 0.0, n, n, n, 2, 2, 2), {}),
        ("wide cross 2-D f32 (radius 2)", ("float32", st, 0.0, 8 * n, 8 * n, 0, 2, 2, 0), {}),
        # radius-2 boxes (125 / 25 points): no fused kernel, one generic launch per operator
        ("big box 3-D f32 (radius 2, 125 points)", ("float32", min(st, 4), 0.0, n, n, n, 2, 2, 2), {"stencil_shape": "box"}),
        ("big box 2-D f32 (radius 2, 25 points)", ("float32", min(st, 8), 0.0, 8 * n, 8 * n, 0, 2, 2, 0), {"stencil_shape": "box"}),
        # extent 3 (round 5: the dense kernel's streaming form in any order of the terms)
        ("cross 3-D f32 (radius 3)", ("float32", min(st, 8), 0.0, n, n, n, 3, 3, 3), {}),
        ("diffusion 3-D f32 (radius 3)", ("float32", min(st, 8), 0.0, n, n, n, 3, 3, 3), {"stencil_shape": "diffusion"}),
        ("cross 3-D f64 (radius 3)", ("float64", min(st, 4), 0.0, n, n, n, 3, 3, 3), {}),
        ("cross 2-D f32 (radius 3)", ("float32", min(st, 8), 0.0, 8 * n, 8 * n, 0, 3, 3, 0), {}),
        ("box 3-D f32 (radius 3, 343 points)", ("float32", 2, 0.0, n, n, n, 3, 3, 3), {"stencil_shape": "box"}),
        ("diffusion 3-D f32 (radius 2)", ("float32", st, 0.0, n, n, n, 2, 2, 2), {"stencil_shape": "diffusion"}),
        ("box 3-D f64", ("float64", st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": "box"}),
        ("hotspot 3-D f64", ("float64", st, 0.0, n, n, n, 1, 1, 1), {"stencil_shape": "hotspot"}),
        ("box 3-D f32, extra field every 2nd stage", ("float32", st, 0.5, n, n, n, 1, 1, 1), {"stencil_shape": "box"}),
    ]
    # fork / join sections (reference bin/synthesize.py:228-253): two branches of two operators every fourth operator
    forks = [
        ("fork 3-D f32", ("float32", st, 0.0, n, n, n, 1, 1, 1), {"fork_frequency": 0.25}),
        ("fork 2-D f32", ("float32", st, 0.0, 8 * n, 8 * n, 0, 1, 1, 0), {"fork_frequency": 0.25}),
        ("fork 3-D f64", ("float64", st, 0.0, n, n, n, 1, 1, 1), {"fork_frequency": 0.25}),
    ]
    if args.fork:
        cases = [(label + " " + o, pos, dict(kw, _opts=o)) for label, pos, kw in forks for o in ("reorder=0", "reorder=1")]
    else:
        cases += forks
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as tmp:
        for label, pos, kw in cases:
            if args.only and args.only not in label:
                continue
            kw = dict(kw)
            case_opts = kw.pop("_opts", "")
            if case_opts:
                args.opts = case_opts
            prog, _ = programs.synthesize(*pos, **kw)
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            chain = sf.KernelChainGraph(path)
            plan = Plan(lower(chain), options=args.opts or None)
            if plan.scalar_names:
                plan.set_scalars([0.1] * len(plan.scalar_names))
            shape = prog["dimensions"]
            dtype = np.float32 if pos[0] == "float32" else np.float64
            plan.upload([rng.random(shape).astype(dtype) for _ in plan.input_names])
            for _ in range(2):
                plan.execute(1)
                plan.synchronize()
            times = []
            for _ in range(3):
                plan.execute(4)
                plan.synchronize()
                times.append(plan.elapsed_ms() / 4)
            ms = float(np.median(times))
            ops = len(prog["program"])
            cells = float(np.prod(shape)) * ops
            bpu = 8.0 if dtype == np.float32 else 16.0
            d = plan.describe()
            kinds = d.count("[star") + d.count("[compact") + d.count("[wide") + d.count("[dense"), d.count("[point]")
            print(json.dumps({"case": label, "opts": args.opts, "dims": shape, "operators": ops, "launches": plan.num_launches,
                              "streaming/point launches": kinds, "ms": round(ms, 3),
                              "Mcells/s": round(cells / ms / 1e3), "frac_8TB": round(cells * bpu / (ms * 1e-3) / 8e12, 3),
                              "first": plan.describe().splitlines()[1].strip()[:110]}), flush=True)
            plan.close()


if __name__ == "__main__":
    main()
