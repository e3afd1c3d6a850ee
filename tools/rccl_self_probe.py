#!/usr/bin/env python3
"""GPU box with ONE GPU: what of the RCCL transport can be exercised there.
A one-rank "nccl" group under a gloo default group (as bench.py builds it), a
collective on a device tensor, and a batched isend/irecv from the rank to itself
on tensors that alias raw device memory the way SlabRunner's buffers do."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stencilflow_amd.distributed import alias_device_buffer  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.cuda.set_device(0)
    g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60))
    x = torch.ones(1024, device="cuda")
    dist.all_reduce(x, group=g)
    torch.cuda.synchronize()
    print("all_reduce on the RCCL group:", float(x.sum()))
    raw = torch.empty(1 << 24, dtype=torch.uint8, device="cuda")
    raw[:] = 7
    t = alias_device_buffer(raw.data_ptr(), raw.numel(), 0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        t[:8 << 20] = 3
        ops = [dist.P2POp(dist.irecv, t[8 << 20:], 0, g), dist.P2POp(dist.isend, t[:8 << 20], 0, g)]
        try:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            stream.synchronize()
            print("self send/recv of 8 MiB through RCCL:", int(raw[(8 << 20) + 5]), "(expected 3)")
        except Exception as exc:  # noqa: BLE001
            print("self send/recv not supported here:", type(exc).__name__, str(exc).splitlines()[0][:200])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
