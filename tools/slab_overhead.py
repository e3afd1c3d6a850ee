#!/usr/bin/env python3
"""GPU: cost of driving a slab through SlabRunner (3 launches per group, Python
in the loop) against plan.execute, with an exchanger that moves nothing."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.distributed import SlabRunner  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


class Null:
    reserved_cus = 0  # compute units the overlapped interior launch leaves free

    def start(self, tensor, regions, key=None):
        return 1

    def finish(self, handle):
        pass


def main():
    stages = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    shape = (1024, 512, 512)  # two slabs of 512 planes; this process is rank 0
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d(shape, stages), os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
    x = np.random.default_rng(0).random((512, 512, 512), dtype=np.float32)
    for overlap, groups, reserve, early in ((True, 1, 0, False), (True, 2, 0, False), (True, 4, 0, False),
                                            (True, 4, 32, False), (True, 4, 0, True), (True, 4, 32, True),
                                            (True, 8, 32, False), (True, 8, 32, True), (False, 4, 0, False)):
        ex = Null()
        ex.reserved_cus = reserve
        r = SlabRunner(sfir, shape, 0, 2, exchanger=ex, overlap=overlap, groups_per_exchange=groups,
                       early_exchange=early)
        r.upload([x])
        r.execute(); r.synchronize()
        t = time.perf_counter()
        r.execute(); r.synchronize()
        dt = time.perf_counter() - t
        t = time.perf_counter()
        r.execute()
        host = time.perf_counter() - t
        r.synchronize()
        print("slab runner overlap=%s groups/exchange=%d reserved CUs=%d early=%s halo=%d: %.3f ms per chain (%.1f us per group), host enqueue %.3f ms" % (
            overlap, groups, reserve, early, r.halo, dt * 1e3, dt * 1e6 / len(r.steps), host * 1e3))
        r.close()
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d((512, 512, 512), stages), os.path.join(tmp, "p.json"))
        plan = Plan(lower(sf.KernelChainGraph(path)))
    plan.upload([x])
    plan.execute(1); plan.synchronize()
    t = time.perf_counter()
    plan.execute(1); plan.synchronize()
    dt = time.perf_counter() - t
    print("single plan: %.3f ms per chain (%.1f us per group)" % (dt * 1e3, dt * 1e6 / plan.num_launches))


if __name__ == "__main__":
    main()
