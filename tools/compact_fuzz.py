#!/usr/bin/env python3
"""GPU: random chains of COMPACT operators (any subset of the 27 / 9 offsets of the
previous stage, extra streamed fields, star stages in between; kernels/compact3d.h) on
random, awkward 3-D and 2-D domain sizes, each compared bit for bit with the oracle.
Only + - * and selects, so every implementation must agree exactly.  Prints one JSON
line per failing program and a progress line every 20 programs.

usage: compact_fuzz.py [--seeds 300] [--first 0] [--options "fuse=3"]"""
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
from oracle import numpy_oracle as npo  # noqa: E402

from tests.random_programs import compact_program as PROGRAM_FN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--options", default="")
    ap.add_argument("--dump", type=int, default=-1, help="print the program of one seed and exit")
    args = ap.parse_args()
    if args.dump >= 0:
        print(json.dumps(PROGRAM_FN(args.dump), indent=1))
        return
    base = {k: v for k, v in (kv.split("=") for kv in args.options.split(";") if kv)}
    nfail = nstar = nlaunch = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(args.first, args.first + args.seeds):
            prog = PROGRAM_FN(seed)
            rng = np.random.default_rng(seed + 7)
            p = npo.load_program(prog)
            ins, arrays, scal = {}, [], {}
            for name, desc in p["inputs"].items():
                dims = npo._input_dims(p, name)
                if dims:
                    ins[name] = rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(
                        npo._NP[desc["data_type"]])
                else:
                    ins[name] = scal[name] = desc["data"]
            want = npo.run_reference(prog, inputs=ins)
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            chain = sf.KernelChainGraph(path)
            opt = dict(base, fuse=int(rng.integers(1, 5 if len(prog["dimensions"]) == 2 else 4)))
            try:
                plan = Plan(lower(chain), options=opt)
            except Exception as exc:  # noqa: BLE001
                nfail += 1
                print(json.dumps({"seed": seed, "error": str(exc)[:300]}), flush=True)
                continue
            desc = plan.describe()
            nstar += desc.count("[compact")
            if (seed - args.first + 1) % 20 == 0:  # a long run must keep writing
                print("# %d programs, %d failures so far" % (seed - args.first + 1, nfail), flush=True)
            nlaunch += plan.num_launches
            if plan.scalar_names:
                plan.set_scalars([scal[n] for n in plan.scalar_names])
            outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                    for n in plan.output_names]
            plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
            plan.close()
            for n, got in zip(plan.output_names, outs):
                if not np.array_equal(got, want[n], equal_nan=True):
                    nfail += 1
                    bad = np.argwhere(~((got == want[n]) | (np.isnan(got) & np.isnan(want[n]))))
                    print(json.dumps({"seed": seed, "output": n, "opt": opt, "nbad": int(len(bad)),
                                      "first_bad": bad[0].tolist(), "dims": prog["dimensions"],
                                      "maxrel": npo.max_rel_err(want[n], got),
                                      "sched": desc[:600]}), flush=True)
    print("programs: %d, launches: %d (compact: %d), failures: %d" % (args.seeds, nlaunch, nstar, nfail))


if __name__ == "__main__":
    main()
