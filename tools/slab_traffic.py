#!/usr/bin/env python3
"""GPU, one process: the launches rank 0 of an N-rank slab run of the headline chain makes
(bench.py --gpus N: jacobi3d (512 N) x 512 x 512, 1000 operators, 512 planes per rank), with NO
neighbour -- the ghost planes are never refreshed, so the numbers it computes are not a stencil
solution; the code object, the plane ranges of its launches and therefore its HBM traffic are
those of the real run.  The slab kernel is its own code object (the global plane count is a
compile-time constant of the generated source), so the undivided kernel's counters do not apply
to it: tools/profile_round.sh runs this under rocprofv3 (kernel trace, then FETCH_SIZE and
WRITE_SIZE in separate passes) and tools/hbm_traffic.py divides the summed counters by the
full-slab launches printed here -> `roofline.traffic` of the N > 1 bench lines.

usage: slab_traffic.py --world N [--out planes.json] [--size 512] [--stages 1000]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class NoNeighbour:
    """Exchanger stand-in: nothing is sent, nothing arrives."""
    reserved_cus = 0

    def start(self, tensor, regions, key=None):
        return None

    def finish(self, handle):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--out", default="")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--stages", type=int, default=1000)
    ap.add_argument("--groups", type=int, default=8, help="launches per exchange (bench.py: 8)")
    args = ap.parse_args()
    import bench
    from stencilflow_amd.distributed import SlabRunner
    wl = bench.make_workload("c3", args.size, args.stages, args.world)
    _, sfir = bench.lower_program(wl["prog"])
    runner = SlabRunner(sfir, wl["shape"], 0, args.world, device=0, exchanger=NoNeighbour(),
                        groups_per_exchange=args.groups)
    runner.upload([bench.synthetic_planes(runner.lo, runner.hi, wl["shape"][1:])])
    runner.execute()  # untimed, unprofiled: first use of the code object
    runner.synchronize()
    plan = runner.plan
    plan.set_profile(True)
    runner.execute()
    runner.synchronize()
    plan.synchronize()  # (collects the launches' events)
    stats, planes = plan.kernel_stats(), plan.kernel_planes()
    plan.set_profile(False)
    name = max(stats, key=lambda k: stats[k]["algorithmic_bytes_per_launch"])
    rec = {"kernel": name, "world": args.world, "n_local": runner.n_local, "halo": runner.halo,
           "executions": 2, "launches_per_execution": stats[name]["launches"],
           "planes_per_execution": planes[name],
           "full_slab_launches_per_execution": planes[name] / float(runner.n_local),
           "avg_launch_us": stats[name]["total_ms"] * 1e3 / max(1, stats[name]["launches"]),
           "schedule": plan.describe().splitlines()[1].strip()}
    runner.close()
    print(json.dumps(rec), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f)


if __name__ == "__main__":
    main()
