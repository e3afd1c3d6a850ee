#!/usr/bin/env python3
"""Headline benchmark: jacobi3d 512^3 float32, 1000-operator chain (BASELINE.json
configs[2]) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one execution of the whole 1000-operator chain on a synthetic
grid resident in HBM.  N > 1 runs the slab-decomposed configuration configs[3]:
(512*N) x 512 x 512, split along the outermost axis, one rank per GPU, halos
exchanged over RCCL / xGMI -- per-GPU work is fixed, so scaling is weak.  Started
under torch.distributed.run the process is one of the N ranks; started as a plain
command with --gpus N > 1 it launches the N ranks itself (before anything
touches a GPU), relays rank 0's line and fails unless N ranks really took part
(the role of `mpirun -n N bin/run_distributed_program.py` in the reference,
bin/run_distributed_program.py:98-100,283-299).  Rank 0 prints ONE JSON line
(contract: task description).

The chain's plan since round 5: 332 launches of THREE operators each (kernels/dense3d.h, fused streaming form) and the
last four operators two by two on the star kernel; `roofline` is about the former's launches (`roofline.scope`).

`roofline` describes the dominant kernel (on rank 0 for N > 1: per GPU):
  achieved / frac   HBM bytes the kernel's launches move / their duration (HIP
                    events on the plan's stream), against the 8 TB/s peak; the bytes
                    are the rocprofv3 PMC traffic of profiles/hbm_traffic.json when
                    that was measured for exactly this code object (`basis`
                    "pmc"), else the compulsory minimum, one read and one write of
                    the field (`basis` "compulsory") -- never above 1;
  traffic           the PMC bytes per (full-slab) launch, or null without a record;
  algorithmic_*     2 * sizeof(dtype) per cell update x the updates
                    of a launch.  A launch fuses `fused_operators` operators, so
                    this figure counts bytes that never travel and may exceed the
                    peak: it is the chain's speed-up over unfused sweeps, not a
                    bandwidth.  (The more operators a launch fuses, the fewer bytes it
                    moves per update: `frac` falls while `value` rises -- the launch of
                    three sits at 0.55 where the launch of two sat at 0.69.)
At N = 1 the line also carries `cpu_baseline` (the oracle's C/OpenMP port on the
host cores) and `other_configs`: BASELINE.json's configs[1] (jacobi2d 4096^2) and
configs[4] (the fused f64 chain), timed the same way for a few steps each, plus two
workloads of the reference's generator on the other fused kernel families.

N > 1: the halo transport is chosen from a ladder -- the
library's own RCCL rung (ncclSend / ncclRecv issued by libsf_hip.so), its
peer-to-peer DMA pushes, shared host memory, gloo (torch.distributed's RCCL only when
SF_BENCH_TRANSPORT=torch asks for it) -- within a wall-clock budget; a rung counts only if, ON EVERY RANK, a decomposed run of
the chain's first operators equals the rank's local recomputation of its slab from
the global synthetic input bit for bit (stencilflow_amd.distributed.DecompositionCheck);
the same check runs again after the timed region and the process exits non-zero on a
mismatch (`config.verified`).
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

T_PROCESS_START = time.perf_counter()
# compiled kernels are kept in the tree (git-ignored; tools/warm_test_cache.sh fills the cache where there is no GPU, and it
# travels with the snapshot like the built library): a run on a fresh box does not spend its first seconds in hipRTC
os.environ.setdefault("SF_HIP_CACHE_DIR", os.path.join(ROOT, ".sf_cache"))
HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
SEED = 20261003
BLOCK = 64  # planes per seeded block of the synthetic grid


def synthetic_planes(lo, hi, tail, dtype=np.float32):
    """Planes [lo, hi) of the global synthetic grid: uniform random in [0, 1), generated
    in blocks of 64 planes seeded by (SEED, block index) -- any rank can produce any
    plane range of the global input at a cost proportional to the range (timing on
    constant data flatters the clock: cdna_hip_programming.md §5.4 rule 25)."""
    tail = tuple(tail)
    out = np.empty((max(0, hi - lo), ) + tail, dtype=dtype)
    b = lo // BLOCK
    while b * BLOCK < hi:
        block = np.random.default_rng([SEED, b]).random((BLOCK, ) + tail, dtype=dtype)
        s, e = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        out[s - lo:e - lo] = block[s - b * BLOCK:e - b * BLOCK]
        b += 1
    return out


def synthetic(shape, dtype=np.float32):
    """The whole grid of a single-GPU workload (2-D grids: one block)."""
    if len(shape) == 2:
        return np.random.default_rng([SEED, 0]).random(tuple(shape), dtype=dtype)
    return synthetic_planes(0, shape[0], shape[1:], dtype)


def measured_valu_busy(kernel):
    """Average number of waves per SIMD that were inside a vector instruction during this code object's committed
    PMC pass (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD; profiles/hbm_traffic.json), or None.
    A wave can issue one vector instruction per 4 cycles (1.0 per wave); f32 instructions of two waves overlap on a
    SIMD, f64 ones hold the double-precision pipe for 4 cycles -- so ~1.0 saturates an f64-typed kernel (C2) and an
    f32 one can pass it (the dense kernel: 1.34)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            v = json.load(f)[kernel].get("valu_busy")
        return None if v is None else float(v)
    except (OSError, KeyError, ValueError, TypeError):
        return None


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
    Threads = the CPUs this process may really use (affinity mask capped by the cgroup
    quota) -- no probing of other counts (2 s probes landed on 32 threads
    for a 16-CPU quota and the figure wandered 5.3 - 9.5e3 between rounds); a quarter of
    the budget goes to the same sample on ONE thread (`one_thread`)."""
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
    cells_per_op = float(np.prod(shape))
    one_applied, one_dt = rate(1, 0.25 * budget_s)
    applied, dt = rate(cpus, 0.75 * budget_s)
    return {
        "value": cells_per_op * applied / dt / 1e6,
        "unit": "Mcells/s",
        "cores": cpus,
        "kind": "port",
        "one_thread": cells_per_op * one_applied / one_dt / 1e6,
        "sample": "{} operators of the chain on the full {}x{}x{} grid, {:.1f} s, "
                  "gcc -O3 -fopenmp, {} threads (= usable CPUs; {} logical CPUs on the box); one thread: {} operators, "
                  "{:.1f} s".format(applied, shape[0], shape[1], shape[2], dt, cpus, os.cpu_count(), one_applied, one_dt),
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


# --------------------------------------------------------------------------- workloads
def make_workload(name, size, stages, slab_world=1):
    """(program, shape, numpy dtype, dtype name, algorithmic bytes per update, label, stages)."""
    from stencilflow_amd import programs
    if name == "c3":
        n = size or 512
        shape = (n * slab_world, n, n)
        return dict(prog=programs.jacobi3d(shape, stages), shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0,
                    stages=stages, name=name,
                    label=("jacobi3d {}x{}x{} float32, {}-operator chain, constant BC 0.0, "
                           "coefficient 0.16666666").format(shape[0], shape[1], shape[2], stages))
    if name == "c2":
        n = size or 4096
        shape = (n, n)
        return dict(prog=programs.jacobi2d(shape, stages), shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0,
                    stages=stages, name=name,
                    label="jacobi2d {}x{} float32, {}-operator chain, constant BC 0.0".format(n, n, stages))
    if name == "box":
        n = size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, 1, 1, 1, stencil_shape="box")
        return dict(prog=prog, shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0, stages=stages, name=name,
                    label="27-point box {}^3 float32 (bin/synthesize.py -stencil_shape box), {}-operator chain".format(
                        n, stages))
    if name == "wide":
        n = size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, 2, 2, 2)
        return dict(prog=prog, shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0, stages=stages, name=name,
                    label="radius-2 cross {}^3 float32 (bin/synthesize.py, extents 2 2 2), {}-operator chain".format(
                        n, stages))
    if name == "cross3":
        n = size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, 3, 3, 3)
        return dict(prog=prog, shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0, stages=stages, name=name,
                    label="radius-3 cross {}^3 float32 (bin/synthesize.py, extents 3 3 3), {}-operator chain".format(
                        n, stages))
    if name == "dense":
        n = size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, 2, 2, 2, stencil_shape="box")
        return dict(prog=prog, shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0, stages=stages, name=name,
                    label="125-point box {}^3 float32 (bin/synthesize.py -stencil_shape box, extents 2 2 2), "
                          "{}-operator chain".format(n, stages))
    if name == "fork":
        n = size or 512
        shape = (n, n, n)
        prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, 1, 1, 1, fork_frequency=0.25)
        ops = len(prog["program"])
        return dict(prog=prog, shape=shape, np_dtype=np.float32, dtype="f32", bpu=8.0, stages=ops, name=name,
                    label="fork / join chain {}^3 float32 (bin/synthesize.py -fork_frequency 0.25: {} stages, "
                          "{} operators)".format(n, stages, ops))
    if name == "c5":
        n = size or 512
        shape = (n, n, n)
        stages = max(3, stages // 3 * 3)
        return dict(prog=programs.diffusion_advection_laplacian(shape, repeats=stages // 3), shape=shape,
                    np_dtype=np.float64, dtype="f64", bpu=16.0, stages=stages, name=name,
                    label=("diffusion->advection->laplacian {}^3 float64, {} operators "
                           "({} chain applications), constant BC 0.0").format(n, stages, stages // 3))
    raise SystemExit("unknown workload " + name)


def lower_program(prog):
    import stencilflow_amd as sf
    from stencilflow_amd import programs
    from stencilflow_amd.lowering import lower
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "bench.json")))
        return chain, lower(chain)


def roofline_block(name, moved_per_full_launch, traffic, seconds, launches_equivalent, launches, wl, fused, cells_per_launch):
    """The `roofline` object: `moved_per_full_launch` bytes x `launches_equivalent`
    full-slab launches in `seconds` of HIP-event time."""
    avg_s = seconds / max(1e-30, launches_equivalent)
    alg = cells_per_launch * wl["bpu"]  # 2 * sizeof(dtype) per cell update
    return {
        "bound": "hbm",
        "kernel": name,
        "achieved": moved_per_full_launch / avg_s / 1e9,
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": moved_per_full_launch / avg_s / HBM_PEAK,
        "basis": "pmc" if traffic is not None else "compulsory",
        "traffic": traffic,
        "compulsory_bytes_per_launch": moved_per_full_launch if traffic is None else None,
        "avg_launch_us": avg_s * 1e6,
        "launches": launches,
        "fused_operators": fused,
        "algorithmic_bytes_per_launch": alg,
        "algorithmic_achieved": alg / avg_s / 1e9,
        "algorithmic_frac": alg / avg_s / HBM_PEAK,
    }


MALL_PEAK = 6.8e12  # B/s: a 64 MiB field copied back and forth inside the 256 MiB Infinity Cache (profiles/r03_copy_bw_64MiB_pingpong.log)


def refine_roofline(roof, wl):
    """Two workloads whose HBM fraction would mislead.
    C2: the 64 MiB field of jacobi2d 4096^2 never leaves the Infinity Cache between launches -- FETCH_SIZE / WRITE_SIZE count
    what crosses the fabric, not HBM -- so the launch is set against the rate of a ping-pong copy of the same field
    (`bound: "mall"`, measured peak), and the HBM figure is kept beside it as `hbm_frac` for what it is worth.
    C5: SURVEY.md 8(d) defines the figure of the fused f64 chain as 16 B per point per chain application; `frac` is that
    one (the bytes the counters see, 1.05 x more, stay in `traffic` / `pmc_frac`)."""
    if wl.get("name") == "c2":
        roof["hbm_frac"] = roof["frac"]
        roof.update(bound="mall", peak=MALL_PEAK / 1e9, frac=roof["achieved"] * 1e9 / MALL_PEAK,
                    peak_basis="ping-pong copy of a 64 MiB field inside the Infinity Cache, 6.7-6.9 TB/s "
                               "(profiles/r03_copy_bw_64MiB_pingpong.log); HBM sees none of this traffic in steady state")
    elif wl.get("name") == "c5":
        roof["pmc_frac"] = roof["frac"]
        per_application = wl["bpu"] * float(np.prod(wl["shape"]))  # (one launch = one application of the three fused operators)
        avg_s = roof["avg_launch_us"] * 1e-6
        roof.update(frac=per_application / avg_s / HBM_PEAK, achieved=per_application / avg_s / 1e9,
                    basis="algorithmic (SURVEY 8d: 16 B per point per application of the fused chain)")


def launch_spread(plan, name):
    """min / median / max launch time (us) of kernel `name` over ONE extra, untimed chain execution with HIP
    events around every launch (the timed region carries none), or {} when that cannot be had."""
    try:
        plan.set_profile(True)
        plan.execute(1)
        plan.synchronize()
        times = plan.kernel_launch_times().get(name)
        plan.set_profile(False)
        if not times:
            return {}
        return {"min_us": times[0] * 1e3, "median_us": times[1] * 1e3, "max_us": times[2] * 1e3,
                "spread_basis": "HIP events around every launch of one extra, untimed chain execution"}
    except Exception:  # noqa: BLE001 -- a diagnostic, never at the price of the line
        return {}


def program_traffic(plan, shape, bpu):
    """Whole-program view of a plan with several kernels (fork / join programs): per chain execution the bytes
    every launch must move at least (each field it reads once + the field it writes once) and, where
    profiles/hbm_traffic.json holds the PMC record of every kernel involved, the bytes they did move."""
    field = float(np.prod(shape)) * bpu / 2.0
    names = plan.kernel_names()
    need = measured = 0.0
    complete = True
    per_kernel = {}
    for s in range(plan.num_steps):
        passes = len(set(plan.step_inputs(s))) + len(plan.step_outputs(s))  # (a DAG group writes several fields)
        need += passes * field
        per_kernel.setdefault(plan.step_kernel(s), []).append(passes * field)
    for k, needs in per_kernel.items():
        t = measured_traffic(names[k])
        if t is None:
            complete = False
        else:
            measured += t * len(needs)
    return {"compulsory_bytes_per_execution": need, "pmc_bytes_per_execution": measured if complete else None,
            "pmc_over_compulsory": (measured / need) if complete and need else None, "launches": plan.num_steps}


def time_single(wl, options, steps, warmup, device=0):
    """One workload on one GPU: `steps` timed chain executions (barrier-free at N = 1,
    synchronised on both sides), HIP events around every execution on the plan's stream.
    Returns (result fields, plan description line)."""
    import torch
    from stencilflow_amd.backend import Plan
    chain, sfir = lower_program(wl["prog"])
    plan = Plan(sfir, device=device, options=options)
    try:
        scalar_values = [chain.inputs[k]["data"] for k in plan.scalar_names]
        if scalar_values:
            plan.set_scalars(scalar_values)
        data = [synthetic(wl["shape"], wl["np_dtype"]) for _ in plan.input_names]
        plan.upload(data)
        # one untimed execution before the counted warm-up steps: the first pass loads the
        # code objects and sizes the launch queues -- none of which belongs to a step even
        # when the caller asks for --warmup 0.  The grid is uploaded again afterwards so the
        # timed steps start from the same synthetic data.
        plan.execute(1)
        plan.synchronize()
        plan.upload(data)
        for _ in range(warmup):
            plan.execute(1)
        plan.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_ms = []
        for _ in range(steps):
            plan.execute(1)
            plan.synchronize()  # HIP events bracket the launches of this step on the plan's stream
            step_ms.append(plan.elapsed_ms())
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(sum(step_ms))
        stats = plan.kernel_stats()
        launches = plan.num_launches * steps
        # (the plan also lists the tile-shape candidates it compiled and did not take: only launched kernels count)
        launched = {plan.kernel_names()[plan.step_kernel(s)] for s in range(plan.num_steps)}
        name = max(launched, key=lambda k: stats[k]["algorithmic_bytes_per_launch"] * max(1, stats[k]["launches"]))
        fused = wl["stages"] / plan.num_launches  # operators evaluated per launch
        cells = float(np.prod(wl["shape"])) * wl["stages"] * steps
        compulsory = float(np.prod(wl["shape"])) * wl["bpu"]  # the field once in, once out
        traffic = measured_traffic(name)
        kernels = len(launched)
        # launches and cell updates per chain execution, by kernel
        per_exec = {}
        for st in range(plan.num_steps):
            k = plan.kernel_names()[plan.step_kernel(st)]
            per_exec[k] = per_exec.get(k, 0) + 1
        updates = {k: stats[k]["updates_per_launch"] * n for k, n in per_exec.items()}
        share = updates[name] / max(1.0, sum(updates.values()))
        if kernels == 1:
            roof = roofline_block(name, traffic if traffic is not None else compulsory, traffic, kernel_ms * 1e-3,
                                  launches, launches, wl, fused, cells / launches)
            roof["compulsory_bytes_per_launch"] = compulsory
            refine_roofline(roof, wl)
        elif share >= 0.9 and len(set(plan.step_inputs(0))) == 1:
            # one kernel does (nearly) all the work -- the benchmark's chain of 1000: 332 launches of three operators and
            # the last four operators two by two on the star kernel: the roofline of THAT kernel's launches.  Its time in
            # the timed region = the region's HIP-event time minus the other kernels' launches at their median duration
            # in one extra, untimed execution with events around every launch.
            plan.set_profile(True)
            plan.execute(1)
            plan.synchronize()
            medians = plan.kernel_launch_times()
            plan.set_profile(False)
            others_ms = sum(medians[k][1] * n for k, n in per_exec.items() if k != name)
            own_s = max(1e-9, kernel_ms - steps * others_ms) * 1e-3
            own_launches = per_exec[name] * steps
            fused_own = stats[name]["updates_per_launch"] / float(np.prod(wl["shape"]))
            roof = roofline_block(name, traffic if traffic is not None else compulsory, traffic, own_s, own_launches,
                                  own_launches, wl, fused_own, stats[name]["updates_per_launch"])
            roof["compulsory_bytes_per_launch"] = compulsory
            roof["scope"] = ("the {} launches of `kernel` per chain execution ({:.1%} of the cell updates); the other {} launches "
                             "({}) at their median duration are taken out of the timed region's {:.3f} ms per execution").format(
                                 per_exec[name], share, plan.num_launches - per_exec[name],
                                 ", ".join("{} x {}".format(n, k) for k, n in per_exec.items() if k != name), kernel_ms / steps)
            refine_roofline(roof, wl)
        else:
            # several kernels (fork / join programs): the roofline of the whole execution -- every launch's bytes
            # over the summed HIP-event time of the executions
            prog = program_traffic(plan, wl["shape"], wl["bpu"])
            moved = prog["pmc_bytes_per_execution"] or prog["compulsory_bytes_per_execution"]
            roof = roofline_block(name, moved, prog["pmc_bytes_per_execution"], kernel_ms * 1e-3, steps, launches, wl,
                                  fused, cells / steps)
            roof["compulsory_bytes_per_launch"] = None
            roof["scope"] = "whole chain execution ({} launches of {} kernels); `kernel` is the one with most work".format(
                plan.num_launches, kernels)
            roof["program"] = prog
        roof["valu_waves_active_per_simd"] = measured_valu_busy(name)
        roof.update(launch_spread(plan, name))
        med = float(np.median(step_ms))
        return {"value": cells / elapsed / 1e6, "ms_per_step": elapsed / steps * 1e3,
                "median_ms_per_step": med, "value_at_median": cells / steps / (med * 1e-3) / 1e6 if med > 0 else None,
                "roofline": roof, "schedule": plan.describe().splitlines()[1].strip(), "compiler": plan.compiler()}
    finally:
        plan.close()


# --------------------------------------------------------------------------- N > 1
class Decomposed:
    """The slab-decomposed run of one rank: transport ladder, schedule, check, timing."""

    # torch.distributed's RCCL ("torch") is taken only when asked for (SF_BENCH_TRANSPORT=torch): its process
    # group ends the process from a watchdog thread on an asynchronous error, which no ladder can catch, and the
    # library issues the same RCCL calls itself on the first rung
    RUNGS = ["rccl", "p2p", "shm", "gloo"]
    AFTER_TORCH = ["torch", "shm", "gloo"]
    NAMES = {"rccl": "RCCL send/recv issued by libsf_hip.so (sf_halo_use_rccl: grouped ncclSend / ncclRecv on the "
                     "transport's stream)",
             "p2p": "DMA pushes into the neighbours' ghost planes (HIP IPC, flags in shared host memory; sf_halo_*)",
             "torch": "RCCL send/recv through torch.distributed on device buffers",
             "shm": "pinned host memory shared by the ranks, stream-ordered flags",
             "gloo": "gloo through pinned host buffers"}

    def __init__(self, args, wl, sfir, options, rank, world, local_rank, self_loop):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.args, self.wl, self.sfir, self.options = args, wl, sfir, options
        self.rank, self.world, self.local_rank, self.self_loop = rank, world, local_rank, self_loop
        self.slab_rank, self.slab_world = (1, 3) if self_loop else (rank, world)
        self.shape = wl["shape"]
        self.notes = []
        self.groups_made = {}
        self.instances = 0
        session = [None]
        if rank == 0:
            session[0] = "{}_{:06x}".format(os.getpid(), int.from_bytes(os.urandom(3), "little"))
        dist.broadcast_object_list(session, src=0)
        self.session = session[0]
        # the check program: the chain's first K operators (K planes of dependency cone per side)
        n_min = self.shape[0] // self.slab_world
        self.check_ops = max(2, min(64, args.stages, n_min // 2))
        from stencilflow_amd import programs
        _, self.check_sfir = lower_program(programs.jacobi3d(self.shape, self.check_ops))

    # ---- small collectives over the control plane (gloo)
    def agreed_max(self, *values):
        t = self.torch.tensor(values, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def everywhere(self, ok):
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def planes_of(self, lo, hi):
        # (self-loop test: halos come back to the sender, there is no global grid -- see check())
        return synthetic_planes(lo, hi, self.shape[1:], self.wl["np_dtype"])

    # ---- transports
    def make_exchanger(self, rung):
        """A fresh exchanger of this rung (collective: every rank makes the same calls)."""
        import datetime
        from stencilflow_amd.distributed import PeerExchanger, ShmExchanger, TorchDistExchanger
        self.instances += 1
        tag = "{}_{}".format(self.session, self.instances)
        if rung in ("rccl", "p2p"):
            if self.self_loop and rung == "p2p":
                raise RuntimeError("the peer-to-peer transport has no self-loop mode")
            return PeerExchanger(self.slab_rank, self.slab_world, tag, device=self.local_rank, transport=rung,
                                 self_loop=self.self_loop)
        if rung == "torch":
            if "nccl" not in self.groups_made:
                self.groups_made["nccl"] = self.dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60))
            ex = TorchDistExchanger(self.slab_rank, self.slab_world, group=self.groups_made["nccl"], staging="device",
                                    self_loop=self.self_loop)
            ex.handshake(self.torch.device("cuda", self.local_rank))
            return ex
        if self.self_loop and rung == "shm":
            raise RuntimeError("the shared-memory transport has no self-loop mode")
        if rung == "shm":
            ex = ShmExchanger(self.rank, self.world, tag, device=self.local_rank)
        else:
            ex = TorchDistExchanger(self.slab_rank, self.slab_world, staging="host", self_loop=self.self_loop)
        ex.handshake()
        return ex

    @staticmethod
    def close_exchanger(ex):
        if ex is not None and hasattr(ex, "close"):
            ex.close()

    def make_runner(self, sfir, rung, groups, data=True):
        """(runner, exchanger) of this rung; the runner holds this rank's slab of the global input."""
        from stencilflow_amd.distributed import SlabRunner
        ex = self.make_exchanger(rung)
        try:
            r = SlabRunner(sfir, self.shape, self.slab_rank, self.slab_world, device=self.local_rank,
                           options=self.options, exchanger=ex, groups_per_exchange=groups)
        except Exception:
            self.close_exchanger(ex)
            raise
        if data:
            r.upload([self.planes_of(r.lo, r.hi)])
        return r, ex

    def run_chain(self, runner, native):
        if native:
            runner.execute_native()
            runner.synchronize_native()
        else:
            runner.execute()
            runner.synchronize()
        if hasattr(runner.exchanger, "check"):
            runner.exchanger.check()  # a halo wait that timed out must not pass for a result

    def chain_seconds(self, runner, native):
        """Agreed (max over ranks) wall time of one chain execution, after one untimed one."""
        self.run_chain(runner, native)
        self.dist.barrier()
        t0 = time.perf_counter()
        self.run_chain(runner, native)
        return self.agreed_max(time.perf_counter() - t0)[0]

    # ---- the check
    def build_check(self, rung, groups, early, native, reserved):
        """DecompositionCheck over a fresh exchanger of `rung`, with the schedule under test."""
        from stencilflow_amd.distributed import DecompositionCheck
        made = {}
        settings = {"native": native}

        def make(sfir):
            r, ex = self.make_runner(sfir, rung, groups, data=False)
            r.early_exchange = early
            if reserved is not None and hasattr(ex, "reserved_cus"):
                ex.reserved_cus = reserved
            made["ex"] = ex
            return r

        if self.self_loop:
            # one rank playing rank 1 of 3 whose halos come back to itself: there is no global
            # solution to compare with; the check then only exercises the machinery
            class Mirror:
                def __init__(mirror):
                    mirror.runner = make(self.check_sfir)

                def passes(mirror):
                    mirror.runner.upload([self.planes_of(mirror.runner.lo, mirror.runner.hi)])
                    self.run_chain(mirror.runner, settings["native"])
                    return True

                def close(mirror):
                    mirror.runner.close()
            check = Mirror()
        else:
            check = DecompositionCheck(self.check_sfir, self.shape, self.slab_rank, self.slab_world, self.planes_of, make,
                                       lambda r: self.run_chain(r, settings["native"]), device=self.local_rank,
                                       options=self.options, dtype=self.wl["np_dtype"])
        check.exchanger = made.get("ex")
        check.settings = settings
        return check

    def close_check(self, check):
        """Collective (also for a rank whose check was never built)."""
        self.close_exchanger(getattr(check, "exchanger", None))
        self.dist.barrier()  # no rank frees buffers a neighbour still has mapped
        if check is not None:
            check.close()

    def try_rung(self, rung, native):
        """Set the rung up, prove it (check on every rank) and time one chain execution.
        Returns (seconds or None, runner, exchanger, check, message); collective."""
        runner = ex = check = None
        ok, msg = True, ""
        try:
            check = self.build_check(rung, 8, False, native, None)
            ok = check.passes()
            if not ok:
                msg = "decomposed result differs from the local recomputation"
        except Exception as exc:  # noqa: BLE001 -- any transport failure selects the next rung
            ok = False
            msg = "{}: {}".format(type(exc).__name__, str(exc).splitlines()[0][:160] if str(exc) else "")
        if not self.everywhere(ok):
            self.close_check(check)
            return None, None, None, None, msg or "failed on another rank"
        try:
            runner, ex = self.make_runner(self.sfir, rung, 8)
            seconds = self.chain_seconds(runner, native)
        except Exception as exc:  # noqa: BLE001
            ok = False
            msg = "{}: {}".format(type(exc).__name__, str(exc).splitlines()[0][:160] if str(exc) else "")
            seconds = None
        if not self.everywhere(ok):
            self.close_exchanger(ex)
            self.dist.barrier()
            if runner is not None:
                runner.close()
            self.close_check(check)
            return None, None, None, None, msg or "failed on another rank"
        return seconds, runner, ex, check, ""

    def select_transport(self):
        """Walk the ladder within the wall-clock budget; keep the fastest proven rung."""
        first = os.environ.get("SF_BENCH_TRANSPORT")
        if first == "torch-rccl":
            first = "torch"
        budget = float(os.environ.get("SF_BENCH_LADDER_SECONDS", "60"))
        # a pinned rung is taken without probing the others; when it fails the ladder continues below it
        if first == "torch":
            ladder = list(self.AFTER_TORCH)
        else:
            ladder = self.RUNGS[self.RUNGS.index(first):] if first in self.RUNGS else list(self.RUNGS)
        t_begin = time.perf_counter()
        best = None
        self.ladder = []  # one record per rung tried: probe time, outcome, reason (config.ladder of the line)
        for rung in ladder:
            library_rung = rung in ("rccl", "p2p")
            if best is not None:
                (elapsed, ) = self.agreed_max(time.perf_counter() - t_begin)
                if first in self.RUNGS or first == "torch" or not library_rung or elapsed > budget:
                    break
            # the library's transports run the library's schedule (sf_plan_execute_decomposed); the spare rungs below
            # them -- host memory, gloo -- have no sf_halo handle and run SlabRunner's Python form of it
            native = library_rung
            t_rung = time.perf_counter()
            seconds, runner, ex, check, msg = self.try_rung(rung, native)
            (probe_s, ) = self.agreed_max(time.perf_counter() - t_rung)
            self.ladder.append({"rung": rung, "ok": seconds is not None, "probe_s": round(probe_s, 3),
                                "chain_ms": None if seconds is None else round(seconds * 1e3, 3), "reason": msg or None})
            if seconds is None:
                self.notes.append("{} not used: {}".format(rung, msg))
                continue
            self.notes.append("{}: verified, one chain execution {:.2f} ms".format(rung, seconds * 1e3))
            cand = dict(rung=rung, seconds=seconds, runner=runner, ex=ex, check=check, native=native)
            if best is None or seconds < best["seconds"]:
                best, cand = cand, best
            if cand is not None:
                self.close_exchanger(cand["ex"])
                self.dist.barrier()
                cand["runner"].close()
                self.close_check(cand["check"])
        if best is None:
            raise SystemExit("no halo transport works: " + "; ".join(self.notes))
        return best

    def tune_schedule(self, best):
        """Halo depth, early exchange, reserved units and native / Python schedule, by
        measurement, alike on all ranks (maxima over ranks)."""
        rung, runner, ex = best["rung"], best["runner"], best["ex"]
        proven_native = best["native"]  # what the ladder proved: halo of 8 launches, nothing early, default units
        proven_cus = getattr(ex, "reserved_cus", None)
        deep = runner.halo
        t_deep, t_half = self.agreed_max(runner.measure_exchange(), runner.measure_exchange(depth=max(1, deep // 2)))
        t_launch = best["seconds"] / max(1, len(runner.steps))
        env_groups = os.environ.get("SF_BENCH_GROUPS")
        groups = int(env_groups) if env_groups in ("4", "8") else (8 if t_deep <= 0.85 * t_launch else 4)
        if groups != 8:
            self.close_exchanger(ex)
            self.dist.barrier()
            runner.close()
            runner, ex = self.make_runner(self.sfir, rung, groups)
            best["runner"], best["ex"] = runner, ex
        best["groups"] = groups
        self.notes.append("exchange alone {:.0f} us ({} planes) / {:.0f} us ({} planes), launch group {:.0f} us".format(
            t_deep * 1e6, deep, t_half * 1e6, max(1, deep // 2), t_launch * 1e6))
        # Two refinements of the schedule, decided by timing whole chain executions (an exchange timed
        # alone says nothing about copies that queue behind a compute kernel holding every unit):
        #  - the exchange started a launch ahead (two interiors of cover for a slow transfer, +4 % of
        #    driver overhead on one GPU);
        #  - compute units left free beside an exchange (for RCCL's copy kernels, or for blit kernels
        #    where a DMA push is not served by the copy engines; that launch is ~12 % longer).
        env_early, env_cus = os.environ.get("SF_BENCH_EARLY_EXCHANGE"), os.environ.get("SF_BENCH_RESERVED_CUS")
        earlies = [env_early == "1"] if env_early in ("0", "1") else [False, True]
        device_side = rung in ("rccl", "p2p", "torch")
        cuses = [int(env_cus)] if env_cus is not None else ([0, 32] if device_side else [0])
        timing = {}
        for early in earlies:
            for cus in cuses:
                runner.early_exchange = early
                if hasattr(ex, "reserved_cus"):
                    ex.reserved_cus = cus
                timing[(early, cus)] = self.chain_seconds(runner, best["native"])
        early, cus = min(timing, key=timing.get)
        runner.early_exchange = early
        if hasattr(ex, "reserved_cus"):
            ex.reserved_cus = cus
        self.notes.append("exchange {}, {} units reserved beside it ({})".format(
            "started a launch ahead" if early else "started with the launch that needs it", cus,
            ", ".join("{}/{}: {:.2f} ms".format("ahead" if e else "with", c, v * 1e3) for (e, c), v in timing.items())))
        # the check follows the schedule that will be timed
        if groups == 8:
            check = best["check"]
            check.runner.early_exchange = runner.early_exchange
            check.settings["native"] = best["native"]
            if hasattr(check.exchanger, "reserved_cus"):
                check.exchanger.reserved_cus = getattr(ex, "reserved_cus", 0)
        else:  # another halo depth is another plan
            self.close_check(best["check"])
            best["check"] = None
            best["check"] = self.build_check(rung, groups, runner.early_exchange, best["native"],
                                             getattr(ex, "reserved_cus", None))
        ok = False
        try:
            ok = best["check"].passes() and os.environ.get("SF_BENCH_TEST_REJECT_TUNED") != "1"  # (test hook)
        except Exception as exc:  # noqa: BLE001
            self.notes.append("check of the tuned schedule raised {}: {}".format(type(exc).__name__, exc))
        if self.everywhere(ok):
            return
        # The refinements are speed only: a tuned schedule that does not reproduce the local recomputation is
        # dropped for the one the ladder proved, and that one is proven again before anything is timed.
        self.notes.append("TUNED SCHEDULE REJECTED by the check: back to the schedule the ladder proved")
        self.close_check(best["check"])
        best["check"] = None
        self.close_exchanger(best["ex"])
        self.dist.barrier()
        best["runner"].close()
        best["native"], best["groups"] = proven_native, 8
        best["check"] = self.build_check(rung, 8, False, proven_native, proven_cus)
        ok = False
        try:
            ok = best["check"].passes()
        except Exception as exc:  # noqa: BLE001
            self.notes.append("check of the proven schedule raised {}: {}".format(type(exc).__name__, exc))
        if not self.everywhere(ok):
            raise SystemExit("neither the tuned nor the proven schedule reproduces the local recomputation: " +
                             "; ".join(self.notes))
        runner, ex = self.make_runner(self.sfir, rung, 8)
        runner.early_exchange = False
        if proven_cus is not None and hasattr(ex, "reserved_cus"):
            ex.reserved_cus = proven_cus
        best["runner"], best["ex"] = runner, ex

    def per_gpu_roofline(self, best, fused):
        """One more, untimed chain execution with HIP events around every launch of rank 0's
        plan: the dominant kernel's bytes over its launch time (per GPU)."""
        runner = best["runner"]
        plan = runner.plan
        ex = best["ex"]
        plan.set_profile(True)
        if hasattr(ex, "set_profile"):
            ex.set_profile(True)
        self.run_chain(runner, best["native"])
        plan.synchronize()
        stats, planes = plan.kernel_stats(), plan.kernel_planes()
        plan.set_profile(False)
        self.exchange = None
        if hasattr(ex, "exchange_times"):
            count, mean_ms, max_ms = ex.exchange_times()
            ex.set_profile(False)
            self.exchange = {"exchanges": count, "mean_us": round(mean_ms * 1e3, 1), "max_us": round(max_ms * 1e3, 1)}
        name = max(stats, key=lambda k: stats[k]["algorithmic_bytes_per_launch"])
        n_local = runner.n_local
        full_launches = planes[name] / float(n_local)  # launches over ranges of other lengths, in full-slab units
        plane_cells = float(np.prod(self.shape[1:]))
        if stats[name]["updates_per_launch"] > 0:  # (operators per launch of THIS kernel: the chain's last operators may run two by two)
            fused = round(stats[name]["updates_per_launch"] / (n_local * plane_cells))
        compulsory = n_local * plane_cells * self.wl["bpu"]
        traffic = measured_traffic(name)
        roof = roofline_block(name, traffic if traffic is not None else compulsory, traffic, stats[name]["total_ms"] * 1e-3,
                              full_launches, stats[name]["launches"], self.wl, fused, n_local * plane_cells * fused)
        roof["compulsory_bytes_per_launch"] = compulsory
        if self.exchange:
            # what an exchange has to hide behind: ONE interior launch (two with the exchange started a launch ahead)
            self.exchange["interior_launch_us"] = roof.get("avg_launch_us")
            self.exchange["cover_launches"] = 2 if getattr(runner, "early_exchange", False) else 1
            self.exchange["scope"] = ("rank 0: HIP events on the transport's streams, from the moment the launches an exchange waits "
                                      "for are done to the moment its planes have arrived (sf_halo_exchange_times)")
            roof["exchange"] = self.exchange
        roof["scope"] = ("rank 0, one GPU: {} launches of one chain execution (interior, boundary and halo-extended "
                         "plane ranges) = {:.1f} full-slab launches; bytes scale with the planes a launch writes").format(
                             stats[name]["launches"], full_launches)
        return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c3", "c2", "c5", "box", "wide", "cross3", "dense", "fork"], default="c3",
                    help="c3 = jacobi3d 512^3 f32 (headline, default); c2 = "
                    "jacobi2d 4096^2 f32; c5 = diffusion/advection/laplacian "
                    "512^3 f64; box / wide = the generator's 27-point box chain / radius-2 cross "
                    "chain 512^3 f32 (own records; single GPU only)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--stages", type=int, default=1000)
    ap.add_argument("--options", type=str, default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload: leave out the c2 / c5 lines (`other_configs`)")
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

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # test hook: all ranks on device 0 (exercises the multi-rank path on a one-GPU box;
    # never used for records)
    one_device = os.environ.get("SF_BENCH_SINGLE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # test hook for a one-GPU box: this single process plays rank 1 of 3 (two
    # neighbours) of the decomposed run and every halo it sends comes back to
    # itself -- through the same transport ladder, RCCL first (the data are not a
    # stencil solution; never used for records, the JSON line says so)
    self_loop = world == 1 and os.environ.get("SF_BENCH_SELF_LOOP") == "1"
    slab_world = 3 if self_loop else world
    multi = slab_world > 1

    if args.workload != "c3" and world > 1:
        raise SystemExit("only the c3 workload is slab-decomposed by bench.py")
    wl = make_workload(args.workload, args.size, args.stages, slab_world)
    args.stages = wl["stages"]
    options = {k: v for k, v in (kv.split("=") for kv in args.options.split(";") if kv)}
    metric = ("Mcells/s (updates) and achieved HBM GB/s vs roofline, jacobi3d 512^3 f32" if args.workload == "c3" else
              "Mcells/s (updates) and achieved HBM GB/s vs roofline, " + args.workload)
    result = {
        "metric": metric, "value": None, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": wl["dtype"],
        "data": "synthetic (uniform random [0,1), 64-plane blocks seeded (%d, block))" % SEED,
        "config": {"workload": wl["label"]},
    }

    if not multi:
        timed = time_single(wl, options, args.steps, args.warmup, device=local_rank)
        result["value"], result["ms_per_step"] = timed["value"], timed["ms_per_step"]
        result["roofline"] = timed["roofline"]
        result["config"].update(decomposition="single", ranks=1, transport="none (single GPU)",
                                schedule=timed["schedule"], compiler=timed["compiler"])
        result["median_ms_per_step"], result["value_at_median"] = timed["median_ms_per_step"], timed["value_at_median"]
        if args.workload == "c3" and not args.size and not args.options and not args.no_other_configs:
            # BASELINE.json configs[1] and configs[4] in the driver's own line:
            # a few steps each (about 25 ms and 45 ms of GPU time per step; a step of c5 is 100
            # applications of the fused chain back to back -- a lone 0.4-ms launch between two
            # synchronisations runs 20 % slower than the same launch in a stream of launches)
            # ... and two workloads of the reference's generator (bin/synthesize.py) that exercise the other
            # fused kernel families: the 27-point box chain (dense3d.h, two per launch since round 4; compact3d.h before) and the radius-2 cross chain
            # (wstar3d.h), 16 operators each
            others = []
            # (>= 10 timed steps each and the median beside the mean: SURVEY.md 8d; the whole set costs about
            # 1.5 s of GPU time)
            # (the generator's chains are 4-28 launches long: 30 timed steps behind 5 untimed ones, so that the mean is the
            #  rate a long chain sees -- the clock the chip holds under a vector-bound launch settles over some tens of
            #  launches, profiles/r05_dense_whatif.log -- and not the rate of the first launches after a pause)
            for name, stages, steps, warm in (("c2", 1000, 10, 1), ("c5", 300, 10, 1), ("box", 16, 30, 5), ("wide", 16, 30, 5),
                                              ("cross3", 8, 30, 5), ("dense", 4, 30, 5), ("fork", 16, 30, 5)):
                try:
                    owl = make_workload(name, 0, stages)
                    t = time_single(owl, {}, steps, warm, device=local_rank)
                except Exception as exc:  # noqa: BLE001 -- a side line must not cost the headline
                    others.append({"workload": name, "error": "{}: {}".format(type(exc).__name__, str(exc)[:200])})
                    continue
                others.append({"workload": owl["label"], "value": t["value"], "unit": "Mcells/s", "steps": steps,
                               "ms_per_step": t["ms_per_step"], "median_ms_per_step": t["median_ms_per_step"],
                               "value_at_median": t["value_at_median"], "dtype": owl["dtype"], "roofline": t["roofline"],
                               "schedule": t["schedule"]})
            result["other_configs"] = others
        if not args.no_cpu_baseline and args.workload == "c3":
            try:
                result["cpu_baseline"] = cpu_baseline(wl["shape"], budget_s=args.cpu_seconds)
            except Exception as exc:  # noqa: BLE001 -- reported, never at the price of the measured line
                result["cpu_baseline"] = {"value": None, "unit": "Mcells/s", "cores": 0, "kind": "port",
                                          "sample": "not measured: {}: {}".format(type(exc).__name__, str(exc)[:200])}
        print(json.dumps(result), flush=True)
        return

    # ---- N > 1: one rank per GPU; gloo is the control plane (barriers, maxima over ranks)
    import datetime
    import torch.distributed as dist
    # (a rank that dies must not leave the others waiting for the default half hour)
    patience = datetime.timedelta(seconds=float(os.environ.get("SF_BENCH_CONTROL_TIMEOUT", "300")))
    if self_loop:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29657")
        dist.init_process_group("gloo", rank=0, world_size=1, timeout=patience)
    else:
        dist.init_process_group("gloo", timeout=patience)
    _, sfir = lower_program(wl["prog"])

    # what one GPU does with the undivided 512^3 grid, measured on rank 0 in this very
    # process before the decomposed run (the per-GPU yardstick of the N-GPU value)
    undivided = None
    if rank == 0 and not self_loop and os.environ.get("SF_BENCH_NO_UNDIVIDED") != "1":
        try:
            one = make_workload("c3", args.size, args.stages, 1)
            undivided = time_single(one, options, 2, 1, device=local_rank)["value"]
        except Exception as exc:  # noqa: BLE001 -- a yardstick, not the measurement
            print("bench.py: undivided yardstick not measured: {}: {}".format(type(exc).__name__, exc), file=sys.stderr)
    dist.barrier()

    job = Decomposed(args, wl, sfir, options, rank, world, local_rank, self_loop)
    best = job.select_transport()
    job.tune_schedule(best)
    runner, exchanger, native = best["runner"], best["ex"], best["native"]

    def step():
        if native:
            runner.execute_native()
        else:
            runner.execute()

    def sync():
        if native:
            runner.synchronize_native()
        else:
            runner.synchronize()
        if hasattr(exchanger, "check"):
            exchanger.check()  # a halo wait that timed out must not pass for a result
        dist.barrier()

    # one untimed execution (first full-size halos over every connection), fresh data,
    # the counted warm-up steps, then EXACTLY `steps` timed ones between barriers
    step()
    sync()
    runner.upload([job.planes_of(runner.lo, runner.hi)])
    for _ in range(args.warmup):
        step()
    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    torch.cuda.synchronize()
    elapsed = job.agreed_max(time.perf_counter() - t0)[0]

    # untimed: the check again, with the transport and schedule that were just timed
    verified = False
    try:
        verified = best["check"].passes()
    except Exception as exc:  # noqa: BLE001
        job.notes.append("final check raised {}: {}".format(type(exc).__name__, exc))
    verified = job.everywhere(verified)

    # (self-loop test: the one rank present updates its own slab only)
    cells = float(np.prod(runner.local_shape if self_loop else wl["shape"])) * args.stages * args.steps
    result["value"] = cells / elapsed / 1e6
    result["ms_per_step"] = elapsed / args.steps * 1e3
    fused = args.stages / max(1, len(runner.steps))
    # (the profiled execution exchanges halos: every rank runs it; the timed value is already in hand, so
    # a failure here costs the roofline block, not the line)
    try:
        roof = job.per_gpu_roofline(best, fused)
    except Exception as exc:  # noqa: BLE001
        roof = {"bound": "hbm", "error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
    if rank == 0:
        result["roofline"] = roof
        per_gpu = result["value"] / (1 if self_loop else world)
        result["roofline"]["per_gpu_mcells_per_s"] = per_gpu
        result["roofline"]["undivided_mcells_per_s"] = undivided
        result["roofline"]["per_gpu_vs_undivided"] = (per_gpu / undivided) if undivided else None
    transport = job.NAMES[best["rung"]] + " (" + "; ".join(job.notes) + ")"
    if self_loop:
        transport += " -- SELF-LOOP TEST: rank 1 of 3, halos sent to the rank itself"
    result["config"].update(
        decomposition="slab{} (halo {} planes, one exchange per {} launches, {})".format(
            slab_world, runner.halo, runner.halo // max(1, runner.steps[0][1]), transport),
        ranks=world, transport=best["rung"], ladder=job.ladder,
        wall_s=round(time.perf_counter() - T_PROCESS_START, 1),  # (the driver allows 600 s per run)
        schedule="sf_plan_execute_decomposed (libsf_hip.so)" if native else "SlabRunner (Python form of the same schedule)",
        verified=bool(verified),
        check=("first {} operators, decomposed vs each rank's local recomputation of its slab from the global "
               "synthetic input ({} ghost planes per side), bit for bit, before and after the timed region").format(
                   job.check_ops, job.check_ops))
    if rank == 0:
        print(json.dumps(result), flush=True)
    # transports first, plans after a barrier: no rank frees buffers a neighbour has mapped
    job.close_check(best["check"])
    job.close_exchanger(exchanger)
    dist.barrier()
    runner.close()
    dist.destroy_process_group()
    if not verified:
        raise SystemExit("bench.py: the decomposed run does NOT reproduce the local recomputation: " +
                         "; ".join(job.notes))


if __name__ == "__main__":
    main()
