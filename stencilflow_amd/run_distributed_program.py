"""Driver for one program on the GPUs of one node -- the role of the reference's
``bin/run_distributed_program.py`` (:98-100 rank bookkeeping, :283-299 barrier /
run / barrier, :304-341 reference comparison on one rank), with the grid split
into slabs along the outermost dimension instead of the operator chain split
across FPGAs (``split_sdfg``, stencilflow/sdfg_generator.py:782-1000).

Started as a plain command it launches one process per GPU itself (before
anything touches a GPU); started under ``torch.distributed.run`` / ``mpirun``-like
launchers (``RANK`` / ``WORLD_SIZE`` set) the process is one of the ranks.  Every
rank materialises the inputs, keeps its slab, runs ``SlabRunner`` over the
library's peer-to-peer transport (``sf_halo_*``; shared host memory and gloo as
spares), writes its planes of every output to a part file; rank 0 stitches them
into ``results/<name>/<out>.dat`` and -- with ``compare_to_reference`` -- checks
them against the registered CPU checker, exactly as ``run_program`` does.
"""

import importlib
import os
import subprocess
import sys

import numpy as np

from . import helper
from .kernel_chain_graph import KernelChainGraph
from .log_level import LogLevel
from .lowering import lower

# (the package exports the function `run_program` under the module's name)
_single = importlib.import_module(__package__ + ".run_program")


def launch(argv, gpus):
    """Start `gpus` ranks of the command-line driver and wait for them."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin",
                          "run_distributed_program.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def run_distributed_program(stencil_file, mode="hardware", compare_to_reference=False, input_directory=None,
                            generate_input=False, halo=0, repetitions=1, log_level=LogLevel.BASIC,
                            print_result=False, options=None, single_device=False, tolerance=1e-6):
    """One rank of the decomposed run (RANK / LOCAL_RANK / WORLD_SIZE from the
    environment).  Returns 0 when verified, None without a comparison; raises
    ``ValueError("Result mismatch.")`` like ``run_program``."""
    import torch
    import torch.distributed as dist
    from .distributed import PeerExchanger, ShmExchanger, SlabRunner, TorchDistExchanger
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if single_device else int(os.environ.get("LOCAL_RANK", "0"))
    log = _single._Log(log_level if rank == 0 else LogLevel.NO_LOG)
    if mode not in _single._MODES:
        raise ValueError("Unrecognized execution mode: {}".format(mode))
    if compare_to_reference and _single._REFERENCE_BACKEND is None:
        raise RuntimeError("compare_to_reference needs a CPU checker: register one with "
                           "stencilflow_amd.run_program.set_reference_backend()")
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP backend cannot run without a GPU")
    torch.cuda.set_device(device)
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo")
    try:
        description = helper.parse_json(stencil_file)
        name = _single._program_name(stencil_file)
        log("Creating kernel graph...")
        chain = KernelChainGraph(path=stencil_file, log_level=LogLevel.NO_LOG)
        sfir = lower(chain)
        shape = tuple(description["dimensions"])
        if shape[0] < 2 * world:
            raise ValueError("the outermost dimension ({}) is too short for {} slabs".format(shape[0], world))
        directory = input_directory if input_directory is not None else os.path.dirname(stencil_file)
        inputs = _single._materialise_inputs(chain, description, directory, generate_input)

        session = [None]
        if rank == 0:
            # (a token of this run: a page left in /dev/shm by a run that died cannot collide)
            session[0] = "rdp{}_{:06x}".format(os.getpid(), int.from_bytes(os.urandom(3), "little"))
        dist.broadcast_object_list(session, src=0)

        def everywhere(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        # the first transport that proves itself on every rank: the library's RCCL rung
        # and its peer-to-peer pushes (both verified when the runner attaches them; their
        # set-up is collective-safe: a rank whose local part fails still takes part and
        # all ranks raise), shared host memory, gloo
        opts = dict(kv.split("=") for kv in options.split(";") if kv) if isinstance(options, str) else dict(options or {})
        runner, used = None, None
        for label in ("rccl", "p2p", "shm", "gloo"):
            exchanger, ok = None, True
            try:
                if label in ("rccl", "p2p"):
                    exchanger = PeerExchanger(rank, world, session[0] + label, device=device, transport=label)
                elif label == "shm":
                    exchanger = ShmExchanger(rank, world, session[0] + label, device=device)
                    exchanger.handshake()
                else:
                    exchanger = TorchDistExchanger(rank, world, staging="host")
                    exchanger.handshake()
                runner = SlabRunner(sfir, shape, rank, world, device=device, options=opts, exchanger=exchanger)
            except Exception as exc:  # noqa: BLE001 -- a transport that fails anywhere is skipped everywhere
                ok = False
                log("transport {} failed: {}".format(label, str(exc).splitlines()[0] if str(exc) else type(exc).__name__))
            if everywhere(ok):
                used = label
                break
            # transports first, plans after a barrier: no rank frees buffers a neighbour has mapped
            if exchanger is not None and hasattr(exchanger, "close"):
                exchanger.close()
            dist.barrier()
            if runner is not None:
                runner.close()
                runner = None
        if runner is None:
            raise RuntimeError("no halo transport works on this node")
        log("Running {} on {} slab(s) of {} planes, halos over {}...".format(name, world, runner.n_local, used))
        log(runner.plan.describe(), LogLevel.MODERATE)

        own_dim = helper.ITERATORS[3 - chain.kernel_dimensions]
        local = []
        for in_name in runner.plan.input_names:
            value = inputs[in_name]
            split = chain.inputs[in_name]["input_dims"][:1] == [own_dim]
            local.append(np.ascontiguousarray(value[runner.lo:runner.hi]) if split else value)
        if runner.plan.scalar_names:
            runner.plan.set_scalars([float(inputs[n]) for n in runner.plan.scalar_names])
        outputs_local = [np.zeros(runner.local_shape, dtype=description["program"][n]["data_type"].type)
                         for n in runner.plan.output_names]
        for _ in range(max(1, repetitions)):
            runner.upload(local)
            runner.execute()
            runner.synchronize()
        if hasattr(runner.exchanger, "check"):
            runner.exchanger.check()
        runner.download(outputs_local)

        folder = os.path.join("results", name)
        parts = os.path.join(folder, ".parts")
        os.makedirs(parts, exist_ok=True)
        for out_name, part in zip(runner.plan.output_names, outputs_local):
            part.tofile(os.path.join(parts, "{}.{}".format(out_name, rank)))
        if hasattr(runner.exchanger, "close"):
            runner.exchanger.close()
        dist.barrier()
        runner.close()
        if rank != 0:
            return None

        outputs = {}
        for out_name in description["outputs"]:
            dtype = description["program"][out_name]["data_type"].type
            pieces = [np.fromfile(os.path.join(parts, "{}.{}".format(out_name, r)), dtype) for r in range(world)]
            outputs[out_name] = np.concatenate(pieces).reshape(shape)
            for r in range(world):
                os.remove(os.path.join(parts, "{}.{}".format(out_name, r)))
        os.rmdir(parts)
        _single._dump(outputs, "result", print_result)
        expected = None
        if compare_to_reference:
            log("Executing reference program...")
            expected = _single._REFERENCE_BACKEND(stencil_file, inputs)
        if halo > 0:
            outputs = _single._without_halo(outputs, halo)
            if expected is not None:
                expected = _single._without_halo(expected, halo)
        helper.save_output_arrays(outputs, folder)
        log("Results saved to " + folder)
        if expected is None:
            return None
        reference_folder = os.path.join(folder, "reference")
        os.makedirs(reference_folder, exist_ok=True)
        helper.save_output_arrays(expected, reference_folder)
        log("Comparing to reference...")
        for out_name, got in outputs.items():
            if not helper.arrays_match(np.ravel(expected[out_name]), np.ravel(got), tolerance):
                raise ValueError("Result mismatch.")
        log("Results verified.")
        return 0
    finally:
        if own_group and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def describe_world():
    """(rank, world) of this process as a launcher set them, or None."""
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:
        return int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    return None


__all__ = ["run_distributed_program", "launch", "describe_world"]
