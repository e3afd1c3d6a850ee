#!/usr/bin/env python3
"""Times the host-buffer entry of the C ABI (sf_plan_run = upload + chain +
download) against its device-resident core, for caller-owned NumPy arrays as
run_program hands them over (reference run_program.py:164-178).
usage: host_path_probe.py [--shape 512x512x512] [--stages 8] [--opts "..."]"""
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
    ap.add_argument("--shape", default="512x512x512")
    ap.add_argument("--stages", type=int, default=8)
    ap.add_argument("--opts", default="")
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    shape = tuple(int(v) for v in args.shape.split("x"))
    prog = (programs.jacobi3d if len(shape) == 3 else programs.jacobi2d)(shape, args.stages)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        chain = sf.KernelChainGraph(path)
        sfir = lower(chain)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    out = np.zeros(shape, np.float32)
    ref = np.zeros(shape, np.float32)
    plan = Plan(sfir, options=args.opts or None)
    nbytes = x.nbytes

    def best(fn):
        t = 1e30
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            t = min(t, time.perf_counter() - t0)
        return t

    t_up = best(lambda: plan.upload([x]))
    t_ex = best(lambda: (plan.execute(1), plan.synchronize()))
    t_dn = best(lambda: plan.download([ref]))
    plan.upload([x]); plan.execute(1); plan.synchronize(); plan.download([ref])
    t_run = best(lambda: plan.run([x], [out]))
    print(json.dumps({
        "shape": shape, "stages": args.stages, "field_MiB": nbytes / 2**20,
        "upload_ms": round(t_up * 1e3, 2), "upload_GB/s": round(nbytes / t_up / 1e9, 1),
        "chain_ms": round(t_ex * 1e3, 2),
        "download_ms": round(t_dn * 1e3, 2), "download_GB/s": round(nbytes / t_dn / 1e9, 1),
        "run_ms": round(t_run * 1e3, 2),
        "run_Mcells/s": round(float(np.prod(shape)) * args.stages / t_run / 1e6, 1),
        "resident_Mcells/s": round(float(np.prod(shape)) * args.stages / t_ex / 1e6, 1),
        "check": "same" if np.array_equal(out, ref) else "DIFF"}), flush=True)
    plan.close()


if __name__ == "__main__":
    main()
