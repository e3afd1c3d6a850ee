#!/usr/bin/env python3
"""Headline benchmark: jacobi3d 512^3 float32, 1000-operator chain (BASELINE.json
configs[2]) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one execution of the whole 1000-operator chain on a synthetic
grid resident in HBM.  N > 1 runs the slab-decomposed configuration configs[3]:
(512*N) x 512 x 512, split along the outermost axis, one rank per GPU, halos
exchanged over RCCL -- per-GPU work is fixed, so scaling is weak.  Started under
torch.distributed.run the process is one of the N ranks; started as a plain
command with --gpus N > 1 it launches the N ranks itself (before anything
touches a GPU), relays rank 0's line and fails unless N ranks really took part
(the role of `mpirun -n N bin/run_distributed_program.py` in the reference,
bin/run_distributed_program.py:98-100,283-299).  Rank 0 prints ONE JSON line
(contract: task description; `roofline` and `cpu_baseline` are added at N = 1).

`roofline` (N = 1) describes the dominant kernel:
  achieved / frac   HBM bytes one launch moves / its average duration (HIP events
                    on the plan's stream), against the 8 TB/s peak; the bytes are
                    the rocprofv3 PMC traffic of profiles/hbm_traffic.json when
                    that was measured for exactly this code object (`basis`
                    "pmc"), else the compulsory minimum, one read and one write of
                    the field (`basis` "compulsory") -- never above 1;
  traffic           the PMC bytes per launch, or null without a matching record;
  algorithmic_*     2 * sizeof(dtype) per cell update x the updates
                    of a launch.  A launch fuses `fused_operators` operators, so
                    this figure counts bytes that never travel and may exceed the
                    peak: it is the chain's speed-up over unfused sweeps, not a
                    bandwidth.
"""

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
SEED = 20261003


def synthetic(shape, rank=0, dtype=np.float32):
    """Uniform random grid (timing on constant data flatters the clock:
    cdna_hip_programming.md §5.4 rule 25)."""
    rng = np.random.default_rng(SEED + rank)
    return rng.random(shape, dtype=dtype)


def measured_traffic(kernel):
    """HBM bytes per launch of exactly this code object (kernel name including
    the hash of its generated source) from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE,
    collected in their own --pmc runs of this command by tools/profile_round.sh),
    or None: counters of another code object say nothing about this one."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        return float(table[kernel]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU
    quota (the GPU boxes expose 256 logical CPUs under a 16-CPU quota; 256
    threads there run 10x slower than 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(shape, block=8, budget_s=12.0):
    """The oracle's C/OpenMP restatement timed on the host cores, on a bounded
    sample: the same operator on the same grid, applied `block` operators at a
    time (output fed back as input) until about `budget_s` seconds have passed.
    The thread count is calibrated first (1x and 2x the usable CPUs, 2 s each)."""
    from oracle import c_oracle
    from stencilflow_amd import programs
    ref = c_oracle.CompiledReference(programs.jacobi3d(shape, block))
    out_name = "b{}".format(block - 1)
    x = ref.run({"a": synthetic(shape)})[out_name]  # untimed warm-up

    def rate(threads, seconds):
        ref.threads = threads
        nonlocal x
        applied, t0 = 0, time.perf_counter()
        while True:
            x = ref.run({"a": x})[out_name]
            applied += block
            dt = time.perf_counter() - t0
            if dt >= seconds or applied >= 1000:
                return applied, dt

    cpus = usable_cpus()
    candidates = sorted({max(1, cpus), min(os.cpu_count() or cpus, 2 * cpus)})
    best = max(candidates, key=lambda t: (lambda a, d: a / d)(*rate(t, 2.0)))
    applied, dt = rate(best, budget_s)
    cells = float(np.prod(shape)) * applied
    return {
        "value": cells / dt / 1e6,
        "unit": "Mcells/s",
        "cores": best,
        "kind": "port",
        "sample": "{} operators of the chain on the full {}x{}x{} grid, {:.1f} s, "
                  "gcc -O3 -fopenmp, {} threads ({} usable CPUs of {})".format(
                      applied, shape[0], shape[1], shape[2], dt, best, cpus,
                      os.cpu_count()),
    }


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this
    script under torch.distributed.run (fresh child processes -- the parent never
    initialises a GPU and nothing is re-exec'ed), pass their output through and
    check that the line rank 0 printed reports N ranks.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith("{"):
            line = out
    code = proc.wait()
    if code != 0:
        print("bench.py: the {}-rank launch failed with exit code {}".format(n, code), file=sys.stderr)
        return code
    try:
        result = json.loads(line)
        ranks = result["n_gpus"] if "n_gpus" in result else result["ranks"]
        connected = result.get("config", {}).get("ranks", result.get("ranks"))
    except (TypeError, ValueError, KeyError):
        print("bench.py: the launch printed no result line", file=sys.stderr)
        return 1
    if ranks != n or connected != n:
        print("bench.py: asked for {} ranks, the result line reports n_gpus {} with {} ranks connected".format(
            n, ranks, connected), file=sys.stderr)
        return 1
    return 0


def launch_check(args):
    """--launch-check: the ranks meet on the control plane (gloo) and are counted."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    ranks, world = int(t.item()), dist.get_world_size()
    if dist.get_rank() == 0:
        print(json.dumps({"launch_check": True, "ranks": ranks, "world_size": world,
                          "n_gpus": args.gpus, "config": {"ranks": ranks, "transport": "none (rendezvous only)"}}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ranks == args.gpus else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c3", "c2", "c5", "box"], default="c3",
                    help="c3 = jacobi3d 512^3 f32 (headline, default); c2 = "
                    "jacobi2d 4096^2 f32; c5 = diffusion/advection/laplacian "
                    "512^3 f64; box = the generator's 27-point box chain 512^3 f32 "
                    "(own records; single GPU only)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--stages", type=int, default=1000)
    ap.add_argument("--options", type=str, default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the control group "
                    "(gloo), the ranks are counted and rank 0 prints the count; "
                    "no GPU is touched (tests of the launcher)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only launches the ranks
        # (it has not imported torch or touched a GPU) and relays rank 0's line
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus {} does not match WORLD_SIZE {}".format(args.gpus, world))
    if args.launch_check:
        raise SystemExit(launch_check(args))

    import torch
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # test hook: all ranks on device 0 with the exchange staged through gloo
    # (exercises the multi-rank path on a one-GPU box; never used for records)
    one_device = os.environ.get("SF_BENCH_SINGLE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # test hook for a one-GPU box: this single process plays rank 1 of 3 (two
    # neighbours) of the decomposed run and every halo it sends comes back to
    # itself -- through the same transport ladder, RCCL first (the data are not a
    # stencil solution; never used for records, the JSON line says so)
    self_loop = world == 1 and os.environ.get("SF_BENCH_SELF_LOOP") == "1"
    slab_rank, slab_world = (1, 3) if self_loop else (rank, world)
    multi = slab_world > 1

    if args.workload != "c3" and world > 1:
        raise SystemExit("only the c3 workload is slab-decomposed by bench.py")
    if args.workload == "c3":
        n = args.size or 512
        shape = (n * slab_world, n, n)
        prog = programs.jacobi3d(shape, args.stages)
        np_dtype, dtype_name, bpu = np.float32, "f32", 8.0
        label = ("jacobi3d {}x{}x{} float32, {}-operator chain, constant BC 0.0, "
                 "coefficient 0.16666666").format(shape[0], shape[1], shape[2], args.stages)
    elif args.workload == "c2":
        n = args.size or 4096
        shape = (n, n)
        prog = programs.jacobi2d(shape, args.stages)
        np_dtype, dtype_name, bpu = np.float32, "f32", 8.0
        label = "jacobi2d {}x{} float32, {}-operator chain, constant BC 0.0".format(
            n, n, args.stages)
    elif args.workload == "box":
        n = args.size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", args.stages, 0.0, n, n, n, 1, 1, 1, stencil_shape="box")
        np_dtype, dtype_name, bpu = np.float32, "f32", 8.0
        label = "27-point box {}^3 float32 (bin/synthesize.py -stencil_shape box), {}-operator chain".format(n, args.stages)
    else:
        n = args.size or 512
        shape = (n, n, n)
        args.stages = max(3, args.stages // 3 * 3)
        prog = programs.diffusion_advection_laplacian(shape, repeats=args.stages // 3)
        np_dtype, dtype_name, bpu = np.float64, "f64", 16.0
        label = ("diffusion->advection->laplacian {}^3 float64, {} operators "
                 "({} chain applications), constant BC 0.0").format(n, args.stages, args.stages // 3)
    options = {k: v for k, v in (kv.split("=") for kv in args.options.split(";") if kv)}

    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "bench.json"))
        chain = sf.KernelChainGraph(path)
        sfir = lower(chain)

    transport = None
    if multi:
        import datetime
        from stencilflow_amd.distributed import SlabRunner, TorchDistExchanger
        import torch.distributed as dist
        # gloo is the control plane (barriers, the max over ranks) and the last
        # resort; the halos travel over RCCL (backend "nccl") if a handshake with
        # both neighbours succeeds on EVERY rank, else through pinned host memory
        # shared by the ranks (ShmExchanger: DMA copies, flags raised and awaited
        # by the streams), else through gloo.  SF_BENCH_TRANSPORT=rccl|shm|gloo
        # starts the ladder at that rung (tests).
        from stencilflow_amd.distributed import ShmExchanger
        if self_loop:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29657")
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("gloo")
        first = os.environ.get("SF_BENCH_TRANSPORT", "p2p")
        ladder = ["p2p", "rccl", "shm", "gloo"]
        ladder = ladder[ladder.index(first):] if first in ladder else ladder
        session = [None]
        if rank == 0:
            session[0] = "{}_{}".format(os.getpid(), int(time.time() * 1e3) & 0xffffff)
        dist.broadcast_object_list(session, src=0)
        names = {"p2p": "DMA pushes into the neighbours' ghost planes (HIP IPC, flags in shared host memory; sf_halo_*)",
                 "rccl": "RCCL send/recv on device buffers",
                 "shm": "pinned host memory shared by the ranks, stream-ordered flags",
                 "gloo": "gloo through pinned host buffers"}
        instances = [0]

        def make_exchanger(rung):
            """A fresh exchanger of this rung (collective: every rank makes the same calls)."""
            instances[0] += 1
            tag = "{}_{}".format(session[0], instances[0])
            if rung == "p2p":
                if self_loop:
                    raise RuntimeError("the peer-to-peer transport has no self-loop mode")
                from stencilflow_amd.distributed import PeerExchanger
                return PeerExchanger(rank, world, tag, device=local_rank)
            if rung == "rccl":
                if "rccl" not in groups_made:
                    groups_made["rccl"] = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
                return TorchDistExchanger(slab_rank, slab_world, group=groups_made["rccl"], staging="device",
                                          self_loop=self_loop)
            if rung == "shm":
                if self_loop:
                    raise RuntimeError("the shared-memory transport has no self-loop mode")
                return ShmExchanger(rank, world, tag, device=local_rank)
            return TorchDistExchanger(slab_rank, slab_world, staging="host", self_loop=self_loop)

        groups_made = {}

        def proves_itself(rung):
            """The rung's self-test on every rank: (works everywhere, ranks that passed, message)."""
            ok, msg, candidate, probe = 1, "", None, None
            try:
                candidate = make_exchanger(rung)
                if rung == "p2p":
                    # its proof needs device buffers of a plan: a small decomposed chain,
                    # rank-stamped planes pushed into both neighbours and checked there
                    small = programs.jacobi3d((16 * world, 8, 64), 2)
                    with tempfile.TemporaryDirectory() as tmp2:
                        small_sfir = lower(sf.KernelChainGraph(programs.write_program(small, os.path.join(tmp2, "p.json"))))
                    probe = SlabRunner(small_sfir, (16 * world, 8, 64), rank, world, device=local_rank,
                                       exchanger=candidate, groups_per_exchange=1)
                elif rung == "rccl":
                    candidate.handshake(torch.device("cuda", local_rank))
                else:
                    candidate.handshake()
            except Exception as exc:  # noqa: BLE001 -- any transport failure selects the next rung
                ok = 0
                msg = "{}: {}".format(type(exc).__name__, str(exc).splitlines()[0][:120] if str(exc) else "")
            finally:
                if probe is not None:
                    probe.close()
                if candidate is not None and hasattr(candidate, "close") and rung != "rccl":
                    candidate.close()
            flag = torch.tensor([ok, -ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            count = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(count, op=dist.ReduceOp.SUM)
            return int(flag[0].item()) == 1, int(count.item()), msg

        # device-to-device transports that prove themselves are all kept: the fastest
        # exchange, measured below on the real buffers, wins; the host-staged spares are
        # tried only when none of them works
        working, why, ranks_connected = [], [], 0
        for rung in ladder:
            if working and rung in ("shm", "gloo"):
                break
            ok, connected, msg = proves_itself(rung)
            if ok:
                working.append(rung)
                ranks_connected = connected
                if rung in ("shm", "gloo"):
                    break
            else:
                why.append("{} handshake failed on some rank{}".format(rung, ": " + msg if msg else ""))
        if not working:
            raise SystemExit("no halo transport works: " + "; ".join(why))
        if ranks_connected != world:
            raise SystemExit("{} of {} ranks connected over {}".format(ranks_connected, world, working[0]))
        rung_used = working[0]
        exchanger = make_exchanger(rung_used)
        # The schedule is chosen by measurement, alike on all ranks.  With halos twice
        # as deep an exchange is needed every 8 launches instead of every 4, which
        # saves 2.5 % when real RCCL copy kernels run beside the compute kernel
        # (tools/rccl_overlap_probe.py: 218.4 against 224.1 us per launch group) --
        # if a 16-plane exchange still fits beside ONE interior launch.  If even the
        # 8-plane exchange does not, it is started a launch ahead (two interiors of
        # cover for +4 % of driver overhead, tools/slab_overhead.py).
        def build(groups, ex=None):
            r = SlabRunner(sfir, shape, slab_rank, slab_world, device=local_rank, options=options,
                           exchanger=ex if ex is not None else exchanger, groups_per_exchange=groups)
            r.upload([synthetic(r.local_shape, rank)])
            return r

        def agreed(*values):
            t = torch.tensor(values, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t]

        runner = build(8)
        if len(working) > 1:
            # several device-to-device transports work: run the whole chain with each on
            # the real buffers -- exchanges beside interior launches, as in the timed
            # region -- and keep the faster one (maxima over ranks; an exchange timed
            # alone would not show a transport whose copies queue behind the compute kernel)
            def chain_seconds():
                # (same random data for every candidate: attaching a transport proves
                # itself on the buffers and clears them, and zeros flatter the clock)
                runner.upload([synthetic(runner.local_shape, rank)])
                runner.execute()
                runner.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                runner.execute()
                runner.synchronize()
                return agreed(time.perf_counter() - t0)[0]

            def use(ex):
                # (RCCL's copy kernels want a few free units beside the interior launch)
                if hasattr(ex, "attach"):
                    runner.attach_exchanger(ex)
                else:
                    runner.exchanger = ex

            timing = {rung_used: chain_seconds()}
            for rung in working[1:]:
                other = make_exchanger(rung)
                use(other)
                timing[rung] = chain_seconds()
                if min(timing, key=timing.get) == rung:
                    if hasattr(exchanger, "close") and rung_used != "rccl":
                        exchanger.close()
                    exchanger, rung_used = other, rung
                else:
                    runner.exchanger = exchanger
                    if hasattr(other, "close") and rung != "rccl":
                        other.close()
            why.append("one chain execution: {}".format(", ".join(
                "{} {:.2f} ms".format(k, v * 1e3) for k, v in timing.items())))
        transport = names[rung_used]
        if why:
            transport += " (" + "; ".join(why) + ")"
        deep = runner.halo
        t_deep, t_half = agreed(runner.measure_exchange(), runner.measure_exchange(depth=max(1, deep // 2)))
        runner.execute()
        runner.synchronize()
        t0 = time.perf_counter()
        runner.execute()
        runner.synchronize()
        (t_launch, ) = agreed((time.perf_counter() - t0) / max(1, len(runner.steps)))
        env_groups = os.environ.get("SF_BENCH_GROUPS")
        groups = int(env_groups) if env_groups in ("4", "8") else (8 if t_deep <= 0.85 * t_launch else 4)
        if groups != 8:
            runner.close()
            if hasattr(exchanger, "attach"):  # buffers are registered per plan: a fresh transport
                exchanger.close()
                exchanger = make_exchanger(rung_used)
            runner = build(groups)
        t_exchange = t_deep if groups == 8 else t_half
        env_early = os.environ.get("SF_BENCH_EARLY_EXCHANGE")
        runner.early_exchange = (env_early == "1") if env_early in ("0", "1") else t_exchange > 0.85 * t_launch
        transport += "; exchange alone {:.0f} us ({} planes) / {:.0f} us ({} planes), launch group {:.0f} us -> {}".format(
            t_deep * 1e6, deep, t_half * 1e6, max(1, deep // 2), t_launch * 1e6,
            "started a launch ahead" if runner.early_exchange else "started with the launch that needs it")
        # Compute units left to the exchange's copy kernels by the launch that runs
        # beside them (device-side transports only): with them a copy kernel never
        # waits for a 200-us block to retire, without them that launch is ~12 %
        # shorter.  Which wins depends on how long the transfer occupies its units on
        # the node at hand: both are timed on one chain execution each.
        default_cus = int(getattr(exchanger, "reserved_cus", 0))
        if default_cus > 0 and os.environ.get("SF_BENCH_RESERVED_CUS") is None:
            timing = {}
            for cus in (default_cus, 0):
                exchanger.reserved_cus = cus
                runner.execute()
                runner.synchronize()
                t0 = time.perf_counter()
                runner.execute()
                runner.synchronize()
                (timing[cus], ) = agreed(time.perf_counter() - t0)
            exchanger.reserved_cus = min(timing, key=timing.get)
            transport += "; {} units reserved beside an exchange ({})".format(
                exchanger.reserved_cus, ", ".join("{}: {:.2f} ms".format(k, v * 1e3) for k, v in timing.items()))
        elif default_cus > 0:
            exchanger.reserved_cus = int(os.environ["SF_BENCH_RESERVED_CUS"])
        if self_loop:
            transport += " -- SELF-LOOP TEST: rank 1 of 3, halos sent to the rank itself"
        runner.upload([synthetic(runner.local_shape, rank)])

        def step():
            runner.execute()

        def sync():
            runner.synchronize()
            if hasattr(exchanger, "check"):
                exchanger.check()  # a halo wait that timed out must not pass for a result
            dist.barrier()
    else:
        plan = Plan(sfir, device=local_rank, options=options)
        scalar_values = [chain.inputs[k]["data"] for k in plan.scalar_names]
        if scalar_values:
            plan.set_scalars(scalar_values)
        plan.upload([synthetic(shape, dtype=np_dtype)])

        def step():
            plan.execute(1)

        def sync():
            plan.synchronize()

    # one untimed execution before the counted warm-up steps: the first pass
    # loads the code objects, sizes the launch queues and (N > 1) moves the
    # first full-size halos over every connection -- none of which belongs to a
    # step even when the caller asks for --warmup 0.  The grid is uploaded again
    # afterwards so the timed steps start from the same synthetic data.
    step()
    sync()
    if multi:
        runner.upload([synthetic(runner.local_shape, rank)])
    else:
        plan.upload([synthetic(shape, dtype=np_dtype)])
    for _ in range(args.warmup):
        step()
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        step()
        if not multi:
            # HIP events bracket the launches of this step on the plan's stream
            plan.synchronize()
            kernel_ms += plan.elapsed_ms()
    sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # control plane (gloo)
        elapsed = float(t.item())

    # (self-loop test: the one rank present updates its own slab only)
    cells = float(np.prod(runner.local_shape if self_loop else shape)) * args.stages * args.steps
    result = {
        "metric": "Mcells/s (updates) and achieved HBM GB/s vs roofline, "
                  "jacobi3d 512^3 f32" if args.workload == "c3" else
                  "Mcells/s (updates) and achieved HBM GB/s vs roofline, " + args.workload,
        "value": cells / elapsed / 1e6,
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic (uniform random [0,1), seed %d)" % SEED,
        "config": {
            "workload": label,
            "decomposition": ("slab{} (halo {} planes, one exchange per {} launches, {})".format(
                slab_world, runner.halo, runner.halo // max(1, runner.steps[0][1]), transport)
                if multi else "single"),
            "ranks": ranks_connected if multi else 1,
            "transport": rung_used if multi else "none (single GPU)",
        },
    }
    if not multi:
        stats = plan.kernel_stats()
        launches = plan.num_launches * args.steps
        name = max(stats, key=lambda k: stats[k]["algorithmic_bytes_per_launch"])
        fused = args.stages / plan.num_launches  # operators evaluated per launch
        avg_s = kernel_ms * 1e-3 / launches
        # bytes a launch must move whatever it fuses: the field once in, once out
        compulsory = float(np.prod(shape)) * bpu
        alg = cells * bpu / launches  # 2 * sizeof(dtype) per cell update
        traffic = measured_traffic(name)
        moved = traffic if traffic is not None else compulsory
        result["roofline"] = {
            "bound": "hbm",
            "kernel": name,
            "achieved": moved / avg_s / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": moved / avg_s / HBM_PEAK,
            "basis": "pmc" if traffic is not None else "compulsory",
            "traffic": traffic,
            "compulsory_bytes_per_launch": compulsory,
            "avg_launch_us": avg_s * 1e6,
            "launches": launches,
            "fused_operators": fused,
            "algorithmic_bytes_per_launch": alg,
            "algorithmic_achieved": alg / avg_s / 1e9,
            "algorithmic_frac": alg / avg_s / HBM_PEAK,
        }
        result["config"]["schedule"] = plan.describe().splitlines()[1].strip()
        if not args.no_cpu_baseline and rank == 0 and args.workload == "c3":
            result["cpu_baseline"] = cpu_baseline(shape, budget_s=args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if multi:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
