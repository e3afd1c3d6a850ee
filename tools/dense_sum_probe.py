#!/usr/bin/env python3
"""GPU: the dense kernel's plain-sum form (all rows of a thread accumulated in step, row segments shared)
against its row-by-row form on the generator's boxes of extent 2: time per operator for a list of block
shapes, every result compared bit for bit with the row-by-row form's.
usage: dense_sum_probe.py [--dims 512 512 512] [--dtype float32]"""
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
    ap.add_argument("--dims", type=int, nargs="+", default=[512, 512, 512])
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--stages", type=int, default=2)
    args = ap.parse_args()
    dims = list(args.dims) + [0] * (3 - len(args.dims))
    ext = [2 if d else 0 for d in dims]
    prog, _ = programs.synthesize(args.dtype, args.stages, 0.0, *dims, *ext, stencil_shape="box")
    shape = [d for d in dims if d]
    x = np.random.default_rng(5).uniform(-1, 1, shape).astype(args.dtype)
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json"))))
    cases = [("rows one by one (dense.sum=0)", {"dense.sum": 0}), ("default", {})]
    if len(shape) == 3:
        for bx, by, rj in ((64, 8, 2), (32, 8, 2), (64, 4, 2), (32, 8, 4), (64, 4, 4), (16, 16, 4), (32, 16, 2), (64, 8, 1), (32, 8, 3)):
            cases.append(("sum %dx%dx%d" % (bx, by, rj), {"k1.bx": bx, "k1.by": by, "k1.rj": rj}))
    else:
        for bx in (64, 128, 256):
            cases.append(("sum %d" % bx, {"k2.bx": bx}))
    want = None
    for label, opt in cases:
        try:
            plan = Plan(sfir, options=opt)
        except Exception as exc:  # noqa: BLE001
            print(json.dumps({"case": label, "error": str(exc)[:200]}), flush=True)
            continue
        got = np.zeros(shape, x.dtype)
        plan.run([x], [got], 1)
        plan.upload([x])
        plan.execute(1)
        plan.synchronize()
        t0 = time.perf_counter()
        reps = 5
        plan.execute(reps)
        plan.synchronize()
        ms = (time.perf_counter() - t0) / reps / args.stages * 1e3
        desc = plan.describe().splitlines()[1].strip()
        plan.close()
        if want is None:
            want = got
        print(json.dumps({"case": label, "ms_per_operator": round(ms, 4), "Mcells/s": round(float(np.prod(shape)) / ms / 1e3),
                          "equal": bool(np.array_equal(got, want)), "launch": desc[desc.find("["):][:150]}), flush=True)


if __name__ == "__main__":
    main()
