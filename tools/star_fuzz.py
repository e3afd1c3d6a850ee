#!/usr/bin/env python3
"""GPU: random chains of radius-1 star operators (the shape the fused
plane-streaming kernel handles) on random, awkward domain sizes, each compared
bit for bit with the oracle.  Only + - * and selects, so every implementation
must agree exactly.  Prints one JSON line per failing program.

usage: star_fuzz.py [--seeds 300] [--first 0] [--options "fuse=3"] [--generator star|wide]
(--generator wide: chains of radius-2 stars, kernels/wstar3d.h; fusion depth 1-3;
 --generator dense: operators with dense radius-2 neighbourhoods, kernels/dense3d.h;
 --generator box_sum: chains of plain sums over subsets of {-1,0,1}^d ordered by plane (round 4: the dense kernel's fused
   streaming form, two operators per launch, plan option dense.t2);
 --generator compact: the 27 offsets of radius 1, kernels/compact3d.h (tools/compact_fuzz.py is its own tool);
 --generator dag: forks, joins and intermediates with several readers, kernels/star3d.h's DAG groups (round 4;
   --options "dag.windows=4" lets 3-D programs form them too);
 --copy (or --generator copy = star --copy): a share of the boundary conditions becomes `copy` -- the
   reference's CPU expansion and the oracles have none (stencil/cpu.py:87), so the fused result is
   compared with the library's generic kernel, one operator per launch)"""
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

from tests.random_programs import (box_sum_program, sparse_sum_program, weighted_cross_program, compact_program, dag_program, dense_program, dense_sum_program, star_program,  # noqa: E402
                                   wide_program, with_copy_boundaries)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--options", default="")
    ap.add_argument("--dump", type=int, default=-1, help="print the program of one seed and exit")
    ap.add_argument("--generator", choices=["star", "wide", "dense", "dense_sum", "box_sum", "sparse_sum", "weighted_cross", "compact", "copy", "dag"], default="star")
    ap.add_argument("--seconds", type=float, default=0, help="stop after this many seconds (0: run all seeds)")
    ap.add_argument("--copy", action="store_true",
                    help="turn a share of the boundary conditions into `copy`; reference: the generic kernel")
    args = ap.parse_args()
    if args.generator == "copy":
        args.generator, args.copy = "star", True
    plain = {"wide": wide_program, "dense": dense_program, "dense_sum": dense_sum_program, "box_sum": box_sum_program, "sparse_sum": sparse_sum_program, "weighted_cross": weighted_cross_program, "compact": compact_program,
             "dag": dag_program}.get(args.generator, star_program)
    make = (lambda seed: with_copy_boundaries(plain(seed), seed)) if args.copy else plain
    if args.dump >= 0:
        print(json.dumps(make(args.dump), indent=1))
        return
    base = {k: v for k, v in (kv.split("=") for kv in args.options.split(";") if kv)}
    nfail = nstar = nlaunch = ndone = ndag = 0
    import time
    t_begin = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(args.first, args.first + args.seeds):
            if args.seconds and time.perf_counter() - t_begin > args.seconds:
                break
            ndone += 1
            prog = make(seed)
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
            path = programs.write_program(prog, os.path.join(tmp, "p.json"))
            chain = sf.KernelChainGraph(path)
            if args.copy:
                with Plan(lower(chain), options={"generic_only": 1}) as ref:
                    if ref.scalar_names:
                        ref.set_scalars([scal[n] for n in ref.scalar_names])
                    routs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                             for n in ref.output_names]
                    ref.run([np.ascontiguousarray(ins[n]) for n in ref.input_names], routs, 1)
                    want = dict(zip(ref.output_names, routs))
            else:
                want = npo.run_reference(prog, inputs=ins)
            opt = dict(base, fuse=int(rng.integers(1, 4 if args.generator == "wide" else 5)))
            try:
                plan = Plan(lower(chain), options=opt)
            except Exception as exc:  # noqa: BLE001
                nfail += 1
                print(json.dumps({"seed": seed, "error": str(exc)[:300]}), flush=True)
                continue
            desc = plan.describe()
            nstar += desc.count("[star") + desc.count("[wide star") + desc.count("[dense") + desc.count("[compact")
            ndag += desc.count("[dag:")
            if (seed - args.first + 1) % 25 == 0:  # a long run must keep writing
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
    print("programs: %d, launches: %d (fused kernels: %d, DAG groups: %d), failures: %d" % (ndone, nlaunch, nstar, ndag, nfail))


if __name__ == "__main__":
    main()
