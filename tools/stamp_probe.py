#!/usr/bin/env python3
"""Where a step of the star kernel spends its time (diagnostic build, option
stamp=1): per-wave s_memtime stamps summed over all waves of one chain execution.
usage: stamp_probe.py [--shape 512x512x512] [--stages 8] [--opts "..."]"""
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
    ap.add_argument("--shape", default="512x512x512")
    ap.add_argument("--stages", type=int, default=8)
    ap.add_argument("--opts", default="")
    args = ap.parse_args()
    shape = tuple(int(v) for v in args.shape.split("x"))
    prog = (programs.jacobi3d if len(shape) == 3 else programs.jacobi2d)(shape, args.stages)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
    x = np.random.default_rng(1).random(shape, dtype=np.float32)
    opts = "stamp=1" + (";" + args.opts if args.opts else "")
    plan = Plan(sfir, options=opts)
    plan.upload([x])
    plan.execute(1)
    plan.synchronize()
    plan.debug_counters(5)  # reading clears the counters
    plan.execute(1)
    plan.synchronize()
    d = plan.debug_counters(5)
    total = float(sum(d[:4])) or 1.0
    names = ["publish+barrier", "stage 1 (+wait for the input plane)", "issue of the next loads", "later stages"]
    print(json.dumps({"opts": opts, "ms": round(plan.elapsed_ms(), 3), "waves": d[4],
                      "share": {n: round(v / total, 3) for n, v in zip(names, d[:4])},
                      "ticks_per_wave": {n: round(v / max(1, d[4])) for n, v in zip(names, d[:4])},
                      "sched": plan.describe().splitlines()[1].strip()[:140]}), flush=True)
    plan.close()


if __name__ == "__main__":
    main()
