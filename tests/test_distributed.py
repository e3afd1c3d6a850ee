"""Multi-rank path: slab arithmetic, the neighbour-exchange protocol over
torch.distributed (gloo, world_size 2 and 3, CPU tensors) and the whole
decomposed chain with the oracle standing in for the device (CPU), plus -- on a
GPU box -- two processes sharing the GPU with host-staged exchange."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def test_slab_bounds_and_regions():
    from stencilflow_amd.distributed import halo_regions, slab_bounds
    for n0, world in [(512, 8), (40, 3), (7, 7), (4096, 8)]:
        cuts = [slab_bounds(n0, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n0
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
    r = halo_regions(n_local=10, halo=3, depth=2, plane_bytes=100)
    assert r["send_down"] == (300, 200) and r["recv_down"] == (100, 200)
    assert r["send_up"] == (1100, 200) and r["recv_up"] == (1300, 200)


def _exchange_worker(rank, world, port, depth, halo):
    import torch
    sys.path.insert(0, ROOT)
    from stencilflow_amd.distributed import (TorchDistExchanger, halo_regions)
    _init(rank, world, port)
    n_local, plane = 6 + rank, 16
    buf = torch.zeros((n_local + 2 * halo) * plane, dtype=torch.uint8)
    view = buf.view(n_local + 2 * halo, plane)
    for p in range(n_local):  # owned plane p of rank r holds 10*r + p
        view[halo + p] = 10 * rank + p
    regions = halo_regions(n_local, halo, depth, plane)
    ex = TorchDistExchanger(rank, world)
    ex.finish(ex.start(buf, regions, key=0))
    if rank > 0:
        n_lo = 6 + rank - 1
        for d in range(depth):
            assert int(view[halo - depth + d, 0]) == 10 * (rank - 1) + n_lo - depth + d
    else:
        assert int(view[:halo].max()) == 0
    if rank < world - 1:
        for d in range(depth):
            assert int(view[halo + n_local + d, 0]) == 10 * (rank + 1) + d
    else:
        assert int(view[halo + n_local:].max()) == 0
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_neighbour_exchange_gloo(world):
    _spawn(_exchange_worker, world, 2, 3)


def _chain_worker(rank, world, port, shape, stages, fuse, tmpdir):
    """The SlabRunner protocol with the oracle as the compute stand-in: per
    launch group exchange `fuse` planes, advance the extended slab `fuse`
    operators, keep the owned planes."""
    import torch
    sys.path.insert(0, ROOT)
    from oracle import numpy_oracle as npo
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import (TorchDistExchanger, halo_regions,
                                             slab_bounds)
    _init(rank, world, port)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    lo, hi = slab_bounds(shape[0], rank, world)
    n_local, halo = hi - lo, fuse
    plane_elems = shape[1] * shape[2]
    local = np.zeros((n_local + 2 * halo, ) + tuple(shape[1:]), np.float32)
    local[halo:halo + n_local] = x[lo:hi]
    ex = TorchDistExchanger(rank, world)
    done = 0
    while done < stages:
        t = min(fuse, stages - done)
        buf = torch.from_numpy(local.reshape(-1).view(np.uint8))
        ex.finish(ex.start(buf, halo_regions(n_local, halo, t, plane_elems * 4),
                           key=0))
        g_lo, g_hi = max(0, lo - t), min(shape[0], hi + t)
        ext = local[halo - (lo - g_lo):halo + n_local + (g_hi - hi)]
        sub = programs.jacobi3d(ext.shape, t, bc_value=0.5)
        out = npo.run_reference(sub, {"a": ext})["b%d" % (t - 1)]
        local[halo:halo + n_local] = out[lo - g_lo:lo - g_lo + n_local]
        done += t
    want = npo.run_reference(programs.jacobi3d(shape, stages, bc_value=0.5),
                             {"a": x})["b%d" % (stages - 1)]
    assert np.array_equal(local[halo:halo + n_local], want[lo:hi])
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fuse", [(2, 2), (3, 1), (2, 3)])
def test_decomposed_chain_protocol_gloo(tmp_path, world, fuse):
    _spawn(_chain_worker, world, (18, 6, 8), 5, fuse, str(tmp_path))


def _gpu_worker(rank, world, port, shape, stages, overlap, groups, transport="gloo", early=False, native=False):
    import torch
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from oracle import numpy_oracle as npo
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import PeerExchanger, ShmExchanger, SlabRunner, TorchDistExchanger
    from stencilflow_amd.lowering import lower
    import tempfile
    _init(rank, world, port)
    if transport == "shm":
        exchanger = ShmExchanger(rank, world, "t{}".format(port), device=0)
        exchanger.handshake()
    elif transport == "p2p":
        exchanger = PeerExchanger(rank, world, "t{}".format(port), device=0)  # verified at attach
    else:
        exchanger = TorchDistExchanger(rank, world, staging="host")
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
    runner = SlabRunner(sfir, shape, rank, world, device=0, exchanger=exchanger,
                        overlap=overlap, groups_per_exchange=groups, early_exchange=early)
    runner.upload([x[runner.lo:runner.hi]])
    if native:  # the same schedule run by the library itself (sf_plan_execute_decomposed)
        runner.execute_native()
        runner.plan.synchronize()
    else:
        runner.execute()
        runner.synchronize()
    if transport in ("shm", "p2p"):
        exchanger.check()
    out = np.zeros(runner.local_shape, np.float32)
    runner.download([out])
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    assert np.array_equal(out, want[runner.lo:runner.hi])
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,groups", [(True, 4), (False, 4), (True, 1)])
def test_two_processes_one_gpu_host_staged(overlap, groups):
    """Two ranks (processes) drive their slabs on the same GPU; halos travel
    through gloo.  Everything but the RCCL transport itself is exercised."""
    _spawn(_gpu_worker, 2, (36, 20, 64), 9, overlap, groups)


@pytest.mark.gpu
@pytest.mark.parametrize("world,overlap,groups,early", [(2, True, 4, False), (3, True, 4, True),
                                                        (3, False, 2, False), (2, True, 2, True)])
def test_processes_on_one_gpu_shared_memory_transport(world, overlap, groups, early):
    """The spare transport of bench.py: halos through pinned host memory that the
    ranks share, flags raised and awaited by the streams (ShmExchanger); one of
    three ranks has two neighbours.  Results bit for bit against the oracle."""
    _spawn(_gpu_worker, world, (48, 20, 64), 19, overlap, groups, "shm", early)


@pytest.mark.gpu
@pytest.mark.parametrize("world,overlap,groups,early", [(2, True, 4, False), (3, True, 4, True),
                                                        (3, False, 2, False), (3, True, 1, False)])
def test_processes_on_one_gpu_peer_to_peer_transport(world, overlap, groups, early):
    """The library's own transport (sf_halo_* of the C ABI, PeerExchanger): DMA
    pushes into the neighbour's ghost planes through IPC-mapped device memory,
    flags in shared host memory.  On this one-GPU box the "peers" are processes
    sharing the device -- everything but the xGMI wire; one of three ranks has two
    neighbours.  Results bit for bit against the oracle."""
    _spawn(_gpu_worker, world, (48, 20, 64), 19, overlap, groups, "p2p", early)


@pytest.mark.gpu
@pytest.mark.parametrize("world,groups", [(2, 4), (3, 2), (3, 1)])
def test_native_deep_halo_schedule_matches_the_oracle(world, groups):
    """SlabRunner.execute_native: the library's own schedule (sf_plan_execute_decomposed)
    over its own transport, 2-3 processes on this GPU, against the oracle bit for bit."""
    _spawn(_gpu_worker, world, (48, 20, 64), 19, True, groups, "p2p", False, True)


def _zero_reach_worker(rank, world, port, shape, native):
    """A chain whose third operator reads its input at the point itself only (reach 0 along the slab axis), one
    operator per launch: the launch after it reads ghost planes the zero-reach launch must have recomputed."""
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from oracle import numpy_oracle as npo
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import PeerExchanger, SlabRunner
    from stencilflow_amd.lowering import lower
    import tempfile
    _init(rank, world, port)
    exchanger = PeerExchanger(rank, world, "z{}".format(port), device=0)
    stages = 7
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    for k in (2, 5):  # b2 and b5 become pointwise operators
        prog["program"]["b%d" % k]["computation_string"] = "b{0} = 2.0 * b{1}[i,j,k] + 0.125".format(k, k - 1)
        prog["program"]["b%d" % k]["boundary_conditions"] = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
    x = np.random.default_rng(13).uniform(-1, 1, shape).astype(np.float32)
    # (the generic kernel: a fused star launch reaches one plane whatever its operators read)
    runner = SlabRunner(sfir, shape, rank, world, device=0, exchanger=exchanger, options={"generic_only": 1},
                        groups_per_exchange=4)
    assert [d for _, d in runner.steps] == [1, 1, 0, 1, 1, 0, 1], runner.steps
    runner.upload([x[runner.lo:runner.hi]])
    if native:
        runner.execute_native()
        runner.plan.synchronize()
    else:
        runner.execute()
        runner.synchronize()
    exchanger.check()
    out = np.zeros(runner.local_shape, np.float32)
    runner.download([out])
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    assert np.array_equal(out, want[runner.lo:runner.hi])
    import torch.distributed as dist
    dist.barrier()
    runner.close()
    exchanger.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_zero_reach_launch_inside_a_chain_keeps_the_ghost_planes(native):
    """sf_plan_execute_decomposed used to launch a zero-reach step over the owned planes only and
    leave the count of good ghost planes unchanged; SlabRunner's Python form recomputes them.  Both forms,
    three processes on this GPU, against the oracle bit for bit."""
    _spawn(_zero_reach_worker, 3, (48, 20, 64), native)


def _dag_native_worker(rank, world, port, native):
    """The generator's fork / join program (2-D: DAG groups with two outputs, a join launch reading two slab-split
    fields) on three processes, the library's own schedule and SlabRunner's, against the oracle."""
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from oracle import numpy_oracle as npo
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import PeerExchanger, SlabRunner
    from stencilflow_amd.lowering import lower
    import tempfile
    _init(rank, world, port)
    exchanger = PeerExchanger(rank, world, "d{}".format(port), device=0)
    prog, _ = programs.synthesize("float32", 12, 0.0, 96, 256, 0, 1, 1, 0, fork_frequency=0.25)
    shape = tuple(prog["dimensions"])
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json"))))
    x = np.random.default_rng(17).uniform(-1, 1, shape).astype(np.float32)
    runner = SlabRunner(sfir, shape, rank, world, device=0, exchanger=exchanger, groups_per_exchange=1)
    assert "[dag:" in runner.plan.describe() and not runner.is_chain
    assert any(len(runner.plan.step_outputs(s)) == 2 for s in range(runner.plan.num_steps))
    runner.upload([x[runner.lo:runner.hi]])
    if native:
        runner.execute_native()
        runner.synchronize_native()
    else:
        runner.execute()
        runner.synchronize()
    exchanger.check()
    out = np.zeros(runner.local_shape, np.float32)
    runner.download([out])
    want = npo.run_reference(prog, {"a": x})[prog["outputs"][0]]
    assert np.array_equal(out, want[runner.lo:runner.hi])
    import torch.distributed as dist
    dist.barrier()
    runner.close()
    exchanger.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_dag_groups_across_processes(native):
    """Round 4: launches that materialise several fields under slab decomposition, both forms of the schedule."""
    _spawn(_dag_native_worker, 3, native)


def _c4_worker(rank, world, port, shape, stages, transport, out_dir):
    """One rank of C4's grid: its slab of seeded random data through `stages`
    operators, result written to out_dir/slab<rank>.dat."""
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import PeerExchanger, ShmExchanger, SlabRunner, slab_bounds
    from stencilflow_amd.lowering import lower
    import tempfile
    dist = _init(rank, world, port)
    if transport == "p2p":
        exchanger = PeerExchanger(rank, world, "c4{}".format(port), device=0)
    else:
        exchanger = ShmExchanger(rank, world, "c4{}".format(port), device=0)
        exchanger.handshake()
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d(shape, stages), os.path.join(tmp, "c4.json"))
        sfir = lower(sf.KernelChainGraph(path))
    runner = SlabRunner(sfir, shape, rank, world, device=0, exchanger=exchanger)
    from tests.test_gpu_parity import c4_input
    lo, hi = slab_bounds(shape[0], rank, world)
    runner.upload([c4_input(lo, hi)])
    runner.execute()
    runner.synchronize()
    exchanger.check()
    out = np.empty(runner.local_shape, np.float32)
    runner.download([out])
    out.tofile(os.path.join(out_dir, "slab{}.dat".format(rank)))
    dist.barrier()
    runner.close()
    exchanger.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["p2p", "shm"])
def test_full_c4_grid_across_processes(transport, tmp_path):
    """C4's grid (4096 x 512 x 512 float32) for 120 operators across four processes
    (the box admits six on its one GPU), halos over the library's peer-to-peer
    transport and over the shared-memory spare, deep-halo schedule with overlapped
    exchange: all 1.07 billion results against the C oracle (computed once per session,
    shared with tests/test_gpu_parity.py's eight-slab test).  What a real 8-GPU run adds
    to this is the xGMI wire."""
    from stencilflow_amd.distributed import slab_bounds
    from tests.test_gpu_parity import C4_SHAPE, C4_STAGES, c4_oracle
    shape, world = C4_SHAPE, 4
    _spawn(_c4_worker, world, shape, C4_STAGES, transport, str(tmp_path))
    want = c4_oracle()
    for r in range(world):
        lo, hi = slab_bounds(shape[0], r, world)
        got = np.fromfile(str(tmp_path / "slab{}.dat".format(r)), np.float32).reshape((hi - lo, ) + shape[1:])
        assert np.array_equal(got, want[lo:hi]), (transport, r)


def _handshake_worker(rank, world, port):
    dist = _init(rank, world, port)
    from stencilflow_amd.distributed import TorchDistExchanger
    TorchDistExchanger(rank, world, staging="host").handshake()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_transport_handshake_gloo(world):
    """The exchanger's self-test (bench.py runs it before trusting a transport)."""
    _spawn(_handshake_worker, world)


@pytest.mark.gpu
@pytest.mark.parametrize("first,expect,transport", [(None, "DMA pushes", "p2p"), ("shm", "shared by the ranks", "shm"),
                                                    ("gloo", "gloo", "gloo")])
def test_bench_three_ranks_on_one_gpu_falls_back(first, expect, transport):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one
    rank per process), with all ranks (one of them with two neighbours) pointed
    at this box's only GPU.  RCCL cannot form a communicator there (several ranks, one
    device): the default ladder must agree on that on every rank, go on to the
    library's peer-to-peer pushes (processes sharing a device can map each other's
    buffers), PROVE them -- the decomposed run of the chain's first operators against
    each rank's local recomputation -- and print its line with `verified`; pinned to
    shared host memory or to its last rung (gloo) it uses that one without probing the others."""
    import json
    import subprocess
    env = dict(os.environ, SF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SF_BENCH_TRANSPORT", None)
    if first:
        env["SF_BENCH_TRANSPORT"] = first
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
           "--size", "64", "--stages", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-8000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["scaling"] == "weak" and "slab3" in rec["config"]["decomposition"]
    assert expect in rec["config"]["decomposition"], rec["config"]["decomposition"]
    assert rec["config"]["transport"] == transport and rec["config"]["verified"] is True
    if first is None:
        assert "rccl not used" in rec["config"]["decomposition"]
    assert "192x64x64" in rec["config"]["workload"]
    roof = rec["roofline"]  # per GPU (rank 0), VERDICT r02 next 1a
    assert 0 < roof["frac"] <= 1 and roof["launches"] > 0 and roof["per_gpu_vs_undivided"] > 0


def _corrupted_halo_worker(rank, world, port, corrupt):
    import torch
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import DecompositionCheck, SlabRunner, TorchDistExchanger
    from stencilflow_amd.lowering import lower
    import tempfile
    dist = _init(rank, world, port)
    shape, ops = (48, 20, 64), 12
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(programs.jacobi3d(shape, ops), os.path.join(tmp, "p.json"))))

    def planes_of(lo, hi):
        return np.random.default_rng(3).random(shape, dtype=np.float32)[lo:hi]

    class Corrupting(TorchDistExchanger):
        """The second exchange delivers one wrong value in rank 1's lower ghost planes --
        what a receiver reading stale cache lines would see."""
        exchanges = 0

        def finish(self, handle):
            super().finish(handle)
            if handle is None:
                return
            self.exchanges += 1
            if corrupt and self.rank == 1 and self.exchanges == 2:
                _, _, (tensor, regions, _) = handle
                off, size = regions["recv_down"]
                tensor[off + size // 2:off + size // 2 + 4] = 0x7b

    def make(text):
        return SlabRunner(text, shape, rank, world, device=0, exchanger=Corrupting(rank, world, staging="host"),
                          groups_per_exchange=1)

    def run(runner):
        runner.execute()
        runner.synchronize()

    check = DecompositionCheck(sfir, shape, rank, world, planes_of, make, run, device=0)
    ok = check.passes()
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    # the rank that received the wrong value must see it (its neighbour may too: what rank 1
    # computes from it travels back with the following exchanges)
    assert (not flags[1]) if corrupt else all(flags), flags
    check.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("corrupt", [False, True])
def test_decomposition_check_catches_a_corrupted_halo(corrupt):
    """bench.py's untimed cross-device check (DecompositionCheck: the decomposed run of the
    chain's first operators against each rank's local recomputation of its slab from the
    global input) on two ranks: green on a correct transport, red on the rank whose ghost
    planes received one wrong value."""
    _spawn(_corrupted_halo_worker, 2, corrupt)


def _library_rccl_self_worker(rank, world, port):
    import datetime
    import torch
    sys.path.insert(0, ROOT)
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import PeerExchanger, SlabRunner, TorchDistExchanger
    from stencilflow_amd.lowering import lower
    import tempfile
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    shape, ops = (96, 20, 64), 19
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(programs.jacobi3d(shape, ops, bc_value=0.25),
                                                                os.path.join(tmp, "p.json"))))
    x = np.random.default_rng(21).uniform(-1, 1, (32, ) + shape[1:]).astype(np.float32)
    results = {}
    nccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
    for label, early in (("library", False), ("library-early", True), ("library-python", False), ("torch", False)):
        if label == "torch":
            ex = TorchDistExchanger(1, 3, group=nccl, staging="device", self_loop=True)
        else:
            ex = PeerExchanger(1, 3, "rs{}{}".format(port, label), device=0, transport="rccl", self_loop=True)
        runner = SlabRunner(sfir, shape, 1, 3, device=0, exchanger=ex, groups_per_exchange=2, early_exchange=early)
        runner.upload([x])
        if label in ("library", "library-early"):  # grouped ncclSend / ncclRecv issued by libsf_hip.so, its own schedule
            assert ex._lib.sf_halo_transport(ex._h) == b"rccl"
            runner.execute_native()
            runner.plan.synchronize()
            ex.check()
        else:
            runner.execute()
            runner.synchronize()
        out = np.zeros(runner.local_shape, np.float32)
        runner.download([out])
        results[label] = out
        if hasattr(ex, "close"):
            ex.close()
        runner.close()
    for label in ("library-early", "library-python", "torch"):
        assert np.array_equal(results["library"], results[label]), label
    assert float(np.abs(results["library"]).max()) > 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_library_rccl_rung_sends_to_itself():
    """The RCCL rung of the library's own transport (sf_halo_use_rccl: librccl through
    dlopen, grouped ncclSend / ncclRecv on the transport's stream) on a one-GPU box: a
    communicator of this rank alone, rank 1 of 3 whose halos come back to itself.  The
    library's schedule (plain and with the exchange started a launch ahead), SlabRunner's
    Python form over the same rung, and torch.distributed's RCCL in the same self-loop
    must all give the same planes bit for bit."""
    _spawn(_library_rccl_self_worker, 1)


def _rccl_bound_worker(rank, world, port):
    """An RCCL exchange that makes no progress within the transport's time limit (its stream is held by a flag
    wait this test stages, nothing else is queued): the bounded wait must raise instead of waiting on."""
    import ctypes
    import time
    import torch
    sys.path.insert(0, ROOT)
    os.environ["SF_RCCL_NO_ABORT"] = "1"  # the stall is this rank's own and resolves: nothing to abort
    from stencilflow_amd.distributed import PeerExchanger
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    ex = PeerExchanger(1, 3, "rb{}".format(port), device=0, timeout_ms=1500, transport="rccl", self_loop=True)
    lib = ex._lib
    n_local, halo, plane = 12, 4, 1 << 16
    raw = torch.zeros((n_local + 2 * halo) * plane, dtype=torch.uint8, device="cuda")
    blob = ctypes.create_string_buffer(ex._blob_bytes)
    assert lib.sf_halo_export(ex._h, 0, ctypes.c_void_p(raw.data_ptr()), plane, n_local, halo, blob) == 0
    assert lib.sf_halo_connect(ex._h, 0, None, None) == 0
    flag = torch.zeros(2, dtype=torch.int32, device="cuda")  # flag word (never set) and the wait's status word
    stream = torch.cuda.Stream()
    raw_stream = ctypes.c_void_p(stream.cuda_stream)
    fp, sp = ctypes.c_void_p(flag.data_ptr()), ctypes.c_void_p(flag.data_ptr() + 4)
    assert lib.sf_flag_wait(raw_stream, fp, 1, 5000, sp) == 0  # holds the stream for 5 s
    assert lib.sf_halo_start(ex._h, 0, 4, raw_stream) == 0      # ... and with it the exchange
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="made no progress for 1500 ms"):
        ex.wait_bounded()
    waited = time.perf_counter() - t0
    assert 1.4 <= waited < 4.0, waited
    with pytest.raises(RuntimeError, match="has failed earlier"):
        ex.check()
    stream.synchronize()  # the staged wait gives up after its 5 s; the exchange then runs to itself
    assert int(flag[1].item()) == 1
    ex.close()  # a failed transport is not waited for
    dist.barrier()
    dist.destroy_process_group()


def _rccl_busy_worker(rank, world, port):
    """healthy exchanges behind MORE queued compute than the transport's time limit -- 60 exchanges, each
    behind ~20 ms of kernels on the compute stream, against a limit of 300 ms -- must not cost the communicator: the
    limit bounds the time without progress, and the transport's progress word moves with every exchange."""
    import ctypes
    import time
    import torch
    sys.path.insert(0, ROOT)
    from stencilflow_amd.distributed import PeerExchanger
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    ex = PeerExchanger(1, 3, "rq{}".format(port), device=0, timeout_ms=300, transport="rccl", self_loop=True)
    lib = ex._lib
    n_local, halo, plane = 12, 4, 1 << 16
    raw = torch.zeros((n_local + 2 * halo) * plane, dtype=torch.uint8, device="cuda")
    blob = ctypes.create_string_buffer(ex._blob_bytes)
    assert lib.sf_halo_export(ex._h, 0, ctypes.c_void_p(raw.data_ptr()), plane, n_local, halo, blob) == 0
    assert lib.sf_halo_connect(ex._h, 0, None, None) == 0
    stream = torch.cuda.Stream()
    raw_stream = ctypes.c_void_p(stream.cuda_stream)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(60):
            torch.cuda._sleep(40_000_000)  # ~20 ms of a spinning kernel
            assert lib.sf_halo_start(ex._h, 0, 4, raw_stream) == 0
            assert lib.sf_halo_finish(ex._h, 0, raw_stream) == 0
    queued = time.perf_counter() - t0
    ex.wait_bounded()  # (raises if the communicator was ended)
    total = time.perf_counter() - t0
    stream.synchronize()
    assert total > 0.6 and total > queued, (queued, total)  # the queue really outlasted the 300-ms limit
    ex.check()
    ex.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_rung_does_not_end_a_healthy_communicator_behind_long_compute():
    _spawn(_rccl_busy_worker, 1)


@pytest.mark.gpu
def test_rccl_rung_bounds_a_stalled_exchange():
    """ncclSend / ncclRecv have no time limit; sf_halo_check (PeerExchanger.wait_bounded) waits on
    the host within the transport's limit, fails the transport and never waits for it again."""
    _spawn(_rccl_bound_worker, 1)


def _rccl_self_worker(rank, world, port):
    import datetime
    import torch
    sys.path.insert(0, ROOT)
    from stencilflow_amd.distributed import (TorchDistExchanger, alias_device_buffer, halo_regions)
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
    n_local, halo, depth, plane = 12, 4, 4, 1 << 16
    raw = torch.zeros((n_local + 2 * halo) * plane, dtype=torch.uint8, device="cuda")
    view = raw.view(n_local + 2 * halo, plane)
    for p in range(n_local):
        view[halo + p] = 100 + p
    tensor = alias_device_buffer(raw.data_ptr(), raw.numel(), 0)  # as SlabRunner aliases plan buffers
    ex = TorchDistExchanger(1, 3, group=rccl, staging="device", self_loop=True)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        view[halo] += 1  # queued before the exchange: the transfer must see it
        for _ in range(3):
            ex.finish(ex.start(tensor, halo_regions(n_local, halo, depth, plane), key=0))
        stream.synchronize()
    got = view[:, 0].cpu().tolist()
    # send_down (first owned planes) came back into the lower ghost planes, send_up into the upper
    assert got[:halo] == [101, 101, 102, 103], got
    assert got[halo + n_local:] == [100 + n_local - depth + d for d in range(depth)], got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_exchanger_sends_to_itself():
    """What a one-GPU box can exercise of the RCCL transport: the exchanger's
    batched send/recv on an "nccl" group created under the gloo default group,
    on tensors aliasing raw device memory, ordered against a side stream -- with
    the rank itself as both neighbours (TorchDistExchanger self_loop)."""
    _spawn(_rccl_self_worker, 1)


@pytest.mark.gpu
def test_bench_drops_a_tuned_schedule_the_check_rejects():
    """The refinements of the schedule (halo depth, early exchange, reserved units, native / Python) are
    speed only: when the check rejects the tuned schedule (forced here by a test hook) the run goes back to
    the schedule the ladder proved, proves it again and prints a verified line instead of giving up."""
    import json
    import subprocess
    env = dict(os.environ, SF_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SF_BENCH_TEST_REJECT_TUNED="1")
    env.pop("SF_BENCH_TRANSPORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", "64", "--stages", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-8000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "TUNED SCHEDULE REJECTED" in rec["config"]["decomposition"]
    assert rec["config"]["verified"] is True and rec["value"] > 0 and rec["config"]["transport"] == "p2p"


@pytest.mark.gpu
def test_bench_self_loop_selects_rccl():
    """bench.py's multi-rank path end to end on one GPU with the RCCL rung of the
    transport ladder: one process as rank 1 of 3, every halo sent to the rank
    itself (SF_BENCH_SELF_LOOP=1)."""
    import json
    import subprocess
    env = dict(os.environ, SF_BENCH_SELF_LOOP="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
           "--size", "64", "--stages", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-8000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    deco = rec["config"]["decomposition"]
    assert "slab3" in deco and "RCCL send/recv issued by libsf_hip.so" in deco and "SELF-LOOP TEST" in deco, deco
    assert rec["config"]["transport"] == "rccl" and rec["value"] > 0
    assert 0 < rec["roofline"]["frac"] <= 1


@pytest.mark.gpu
def test_a_communicator_that_does_not_form_in_time_fails_its_rung_only():
    """ncclCommInitRank runs on a helper thread under a bounded wait (PeerExchanger._use_rccl_bounded):
    given no time at all the rung raises -- promptly, instead of holding the run -- and says why; bench.py
    records that on the ladder (`rccl not used: ... did not form within`) and goes on to the next rung
    (none of the others has a self-loop mode, so on this one-GPU box the ladder ends there)."""
    import subprocess
    code = ("import os, sys, time; sys.path.insert(0, {root!r})\n"
            "from stencilflow_amd.distributed import PeerExchanger\n"
            "t0 = time.perf_counter()\n"
            "try:\n"
            "    PeerExchanger(1, 3, 'bounded%d' % os.getpid(), device=0, transport='rccl', self_loop=True)\n"
            "    print('RESULT formed')\n"
            "except RuntimeError as exc:\n"
            "    print('RESULT %.3f %s' % (time.perf_counter() - t0, exc))\n"
            "time.sleep(5)  # (the abandoned call completes in the background)\n").format(root=ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SF_HALO_RCCL_INIT_SECONDS="0.0001")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0]
    assert "did not form within" in line, line
    assert float(line.split()[1]) < 5.0
    env = dict(os.environ, SF_BENCH_SELF_LOOP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SF_HALO_RCCL_INIT_SECONDS="0.0001",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SF_BENCH_TRANSPORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--size", "64", "--stages", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and "rccl not used" in r.stderr and "did not form within" in r.stderr, r.stderr[-4000:]
    assert "p2p not used" in r.stderr  # the ladder went on


@pytest.mark.gpu
def test_program_inputs_no_launch_writes_are_exchanged_once(tmp_path):
    """A chain whose every operator also reads one extra field across planes (the shape of the
    reference generator's extra spatial fields, bin/synthesize.py:170-196) on two slabs: the extra
    field is a program input no launch writes, so its ghost planes are filled once per
    execution -- at the first launch that reads it, to the full halo depth -- and every
    later launch finds them in place; fields the launches produce are exchanged whenever
    they are read across planes.  Result equal to the oracle bit for bit, twice in a row."""
    import numpy as np
    from oracle import numpy_oracle as npo
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from stencilflow_amd.lowering import lower
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": "float32"},
                       "e": {"data": "constant:0.5", "data_type": "float32"}},
            "outputs": ["b3"], "dimensions": [24, 20, 32], "program": {}}
    for t in range(4):
        src = "a" if t == 0 else "b%d" % (t - 1)
        prog["program"]["b%d" % t] = {
            "computation_string": "b{t} = 0.25 * ({s}[i-1,j,k] + {s}[i+1,j,k] + {s}[i,j,k-1] + {s}[i,j+1,k]) + "
                                  "0.125 * (e[i-1,j,k] + e[i+1,j,k] + e[i,j-1,k+1])".format(t=t, s=src),
            "boundary_conditions": {src: {"type": "constant", "value": 0.0}, "e": {"type": "constant", "value": 1.0}},
            "data_type": "float32"}
    rng = np.random.default_rng(77)
    p = npo.load_program(prog)
    ins = {n: rng.uniform(-1, 1, npo._dims_shape(p, npo._input_dims(p, n))).astype(np.float32) for n in p["inputs"]}
    want = npo.run_reference(prog, inputs=ins)
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    shape, world = tuple(prog["dimensions"]), 2
    exch = LocalExchanger(world)
    runners = [SlabRunner(sfir, shape, r, world, options={"fuse": 2}, exchanger=exch.for_rank(r), groups_per_exchange=1)
               for r in range(world)]
    assert runners[0].plan.num_launches >= 2
    counts = [{} for _ in runners]
    for r, seen in zip(runners, counts):
        inner = r.exchanger.start

        def counting(tensor, regions, key=None, _inner=inner, _seen=seen):
            _seen[key] = _seen.get(key, 0) + 1
            return _inner(tensor, regions, key=key)
        r.exchanger.start = counting
    r0 = runners[0]
    fixed = r0._static
    assert fixed and len(fixed) < len(r0.plan.input_names) + 1
    readers = {b: sum(1 for s in range(len(r0.steps)) if b in r0.inputs[s] and r0.steps[s][1] > 0) for b in fixed}
    assert max(readers.values()) >= 2, "the program should read an extra field in more than one launch"
    for execution in range(2):
        for seen in counts:
            seen.clear()
        for r in runners:
            r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi]) for n in r.plan.input_names])
        run_lockstep(runners)
        for seen in counts:
            assert all(seen.get(b, 0) == 1 for b in fixed if readers[b] > 0), (seen, fixed)
            assert any(v > 1 for b, v in seen.items() if b not in fixed) or len(seen) > len(fixed)
        name = r0.plan.output_names[0]
        got = np.zeros(shape, np.float32)
        for r in runners:
            part = [np.zeros(r.local_shape, np.float32) for _ in r.plan.output_names]
            r.download(part)
            got[r.lo:r.hi] = part[0]
        assert np.array_equal(got, want[name])
    for r in runners:
        r.close()


def test_ranks_of_a_decomposed_run_plan_alike(tmp_path):
    """Every rank of a slab decomposition must form the same launch groups on the same kernels (the exchange schedule is
    derived from the plan): the planner's choices -- including round 4's, which look at tile shapes against the grid
    (two boxes per dense launch, compact groups that would mostly recompute, the longer of star chain and compact
    group) -- may depend on the (j,k) extent and on what compiles, never on where a rank's slab lies or how tall it is."""
    import re
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    cases = {
        "box": programs.synthesize("float32", 4, 0.0, 96, 512, 512, 1, 1, 1, stencil_shape="box")[0],
        "box_two_fields": programs.synthesize("float32", 4, 0.5, 96, 512, 512, 1, 1, 1, stencil_shape="box")[0],
        "cross_two_fields": programs.synthesize("float32", 4, 0.5, 96, 512, 512, 1, 1, 1)[0],
    }
    for name, prog in cases.items():
        sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / (name + ".json")))))
        seen = {}
        for slab in (None, "0:24:8:96", "24:56:8:96"):
            with Plan(sfir, options={"slab": slab} if slab else None) as plan:
                launches = re.findall(r"launch (sf_\w+): ([^\[]+)\[", plan.describe())
            seen[slab] = launches
        first = seen[None]
        assert first, name
        for slab, launches in seen.items():
            assert launches == first, (name, slab, launches, first)
