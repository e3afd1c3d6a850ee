#!/usr/bin/env python3
"""One-GPU box: the slab schedule with REAL RCCL send/recv kernels in flight.
A single process is rank 1 of 3 (two neighbours) of a (3*512) x 512 x 512 run;
its exchanger sends every halo to the rank itself (TorchDistExchanger
self_loop), so the copy kernels, their streams and their contention with the
interior launch are the real ones -- only the xGMI wire is missing (the data
move inside HBM).  Results are not a stencil solution; only time is read.

usage: rccl_overlap_probe.py [stages=200]"""
import datetime
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.distributed import SlabRunner, TorchDistExchanger  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


class Null:
    reserved_cus = 0

    def start(self, tensor, regions, key=None):
        return 1

    def finish(self, handle):
        pass


def main():
    stages = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29656")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.cuda.set_device(0)
    rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60))
    shape = (3 * 512, 512, 512)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d(shape, stages), os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
    x = np.random.default_rng(0).random((512, 512, 512), dtype=np.float32)
    cases = [("no transport", None, 0, False, 4), ("no transport, 32 CUs reserved", None, 32, False, 4),
             ("RCCL to self", "rccl", 32, False, 4), ("RCCL to self, no CUs reserved", "rccl", 0, False, 4),
             ("RCCL to self, 8 CUs reserved", "rccl", 8, False, 4),
             ("RCCL to self, 16 CUs reserved", "rccl", 16, False, 4),
             ("RCCL to self, started a launch ahead", "rccl", 32, True, 4),
             ("RCCL to self, exchange per 8 launches", "rccl", 32, False, 8),
             ("RCCL to self, exchange per 8, 16 CUs", "rccl", 16, False, 8),
             ("RCCL to self, exchange per 2 launches", "rccl", 32, False, 2)]
    for label, kind, reserve, early, groups in cases:
        ex = Null() if kind is None else TorchDistExchanger(1, 3, group=rccl, staging="device", self_loop=True)
        ex.reserved_cus = reserve
        r = SlabRunner(sfir, shape, 1, 3, exchanger=ex, early_exchange=early, groups_per_exchange=groups)
        r.upload([x])
        t_ex = r.measure_exchange() if kind else 0.0
        r.execute(); r.synchronize()
        t = time.perf_counter()
        r.execute(); r.synchronize()
        dt = time.perf_counter() - t
        print("%-40s halo %d: %.3f ms per chain, %.1f us per launch group; one exchange alone %.0f us" % (
            label, r.halo, dt * 1e3, dt * 1e6 / len(r.steps), t_ex * 1e6), flush=True)
        r.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
