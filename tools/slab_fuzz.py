#!/usr/bin/env python3
"""GPU: random programs under slab decomposition.  Every seed draws a program
(a random star chain or a random DAG, tests/random_programs.py), a world size,
a halo depth (launch groups per exchange) and the overlap switch, runs all
ranks in this process on one GPU (LocalExchanger copies the halos) and compares
the stitched result bit for bit with the oracle.  Lower-dimensional inputs are
replicated or sliced per rank as the slab runner expects.

usage: slab_fuzz.py [--seeds 100] [--first 0] [--generator mixed|star|wide|compact|dense|box_sum|sparse_sum|weighted_cross] [--copy] [--seconds S]"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402
from oracle import numpy_oracle as npo  # noqa: E402
import tests.random_programs as rp  # noqa: E402
from tests.random_programs import random_program, star_program  # noqa: E402


def run_seed(seed, tmp, generator="mixed", copy=False, seconds_per_case=None):
    """One fuzz case.  Returns ("ok" | "skip" | "fail", detail dict)."""
    rng = np.random.default_rng(seed + 99)
    if generator == "mixed":
        prog = star_program(seed) if rng.random() < 0.7 else random_program(seed)
    else:
        prog = getattr(rp, generator + "_program")(seed)
    if copy:
        prog = rp.with_copy_boundaries(prog, seed)
    dims = prog["dimensions"]
    if len(dims) < 2:
        return "skip", {}
    # make the split axis tall enough for several ranks
    dims[0] = int(rng.integers(24, 64)) if generator in ("mixed", "star") else int(rng.integers(48, 96))
    world = int(rng.integers(2, 5))
    groups = int(rng.choice([1, 2, 4]))
    overlap = bool(rng.random() < 0.6)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        idims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, idims)).astype(npo._NP[desc["data_type"]])
                     if idims else desc["data"])
    path = programs.write_program(prog, os.path.join(tmp, "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
    if copy:
        # the oracles have no `copy` (reference stencil/cpu.py:87 raises): the undivided run on the library's
        # generic kernel, one operator per launch, is the reference
        from stencilflow_amd.backend import Plan
        with Plan(sfir, options={"generic_only": 1}) as ref:
            if ref.scalar_names:
                ref.set_scalars([ins[n] for n in ref.scalar_names])
            routs = [np.zeros(dims, dtype=npo._NP[prog["program"][n]["data_type"]]) for n in ref.output_names]
            ref.run([np.ascontiguousarray(ins[n]) for n in ref.input_names], routs, 1)
            want = dict(zip(ref.output_names, routs))
    else:
        want = npo.run_reference(prog, inputs=ins)
    split = "i" if len(dims) == 3 else "j"  # iterator of the outermost axis
    exch = LocalExchanger(world)
    fuse = int(rng.integers(1, 4))
    if generator in ("wide", "dense", "box_sum", "sparse_sum", "weighted_cross"):
        groups = min(groups, 2)  # (reach 2 per operator: deeper halos than the slabs are tall)
    early = bool(seed % 2)  # exchange started a launch ahead (SlabRunner early_exchange)
    label = {"seed": seed, "world": world, "groups": groups, "overlap": overlap, "dims": dims, "fuse": fuse,
             "early": early}
    try:
        runners = [SlabRunner(sfir, tuple(dims), r, world, options={"fuse": fuse},
                              exchanger=exch.for_rank(r), overlap=overlap, groups_per_exchange=groups,
                              early_exchange=early)
                   for r in range(world)]
    except ValueError as exc:
        if "too thin" in str(exc):
            return "skip", label
        return "fail", dict(label, error=str(exc)[:200])
    plan = runners[0].plan
    try:
        for r in runners:
            local = []
            for name in plan.input_names:
                idims = npo._input_dims(p, name)
                arr = np.ascontiguousarray(ins[name])
                local.append(np.ascontiguousarray(arr[r.lo:r.hi]) if idims and idims[0] == split else arr)
            if plan.scalar_names:
                r.plan.set_scalars([ins[n] for n in r.plan.scalar_names])
            r.upload(local)
        run_lockstep(runners)
        got = {n: np.zeros(dims, dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names}
        for r in runners:
            parts = [np.zeros(r.local_shape, dtype=got[n].dtype) for n in plan.output_names]
            r.download(parts)
            for n, part in zip(plan.output_names, parts):
                got[n][r.lo:r.hi] = part
    except Exception as exc:  # noqa: BLE001
        return "fail", dict(label, error=str(exc)[:200], chain=runners[0].is_chain,
                            halo=[r.halo for r in runners], program=prog)
    finally:
        for r in runners:
            r.close()
    for n in plan.output_names:
        if not np.array_equal(got[n], want[n], equal_nan=True):
            bad = np.argwhere(~((got[n] == want[n]) | (np.isnan(got[n]) & np.isnan(want[n]))))
            return "fail", dict(label, output=n, nbad=int(len(bad)), first_bad=bad[0].tolist(),
                                chain=runners[0].is_chain, program=prog)
    return "ok", label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--generator", choices=["mixed", "star", "wide", "compact", "dense", "box_sum", "sparse_sum", "weighted_cross", "dag"], default="mixed",
                    help="mixed = star chains and random DAGs (the default); the others: tests/random_programs.py")
    ap.add_argument("--copy", action="store_true", help="`copy` boundaries; reference: undivided run on the generic kernel")
    ap.add_argument("--seconds", type=float, default=0, help="stop after this many seconds (0: all seeds)")
    args = ap.parse_args()
    count = {"ok": 0, "skip": 0, "fail": 0}
    import time
    t_begin = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(args.first, args.first + args.seeds):
            if args.seconds and time.perf_counter() - t_begin > args.seconds:
                break
            status, detail = run_seed(seed, tmp, args.generator, args.copy)
            count[status] += 1
            if (seed - args.first + 1) % 20 == 0:  # a long run must keep writing
                print("# %d seeds, %d failures so far" % (seed - args.first + 1, count["fail"]), flush=True)
            if status == "fail":
                print(json.dumps(detail), flush=True)
    print("programs run: %d (skipped %d), failures: %d" % (count["ok"] + count["fail"], count["skip"], count["fail"]))


if __name__ == "__main__":
    main()
