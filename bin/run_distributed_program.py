#!/usr/bin/env python3
"""One program on N GPUs of this node, the grid split into slabs along its outermost
dimension -- the place of the reference's `mpirun -n N bin/run_distributed_program.py`
(bin/run_distributed_program.py:98-100,283-341).  Started plainly it launches its N
ranks itself; under a launcher that sets RANK / WORLD_SIZE it is one of them.

    bin/run_distributed_program.py <program.json> hardware -gpus 4 \\
        [-compare-to-reference -reference-checker module:function] [-halo H] ...
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("stencil_file")
    p.add_argument("mode", nargs="?", default="hardware", choices=["emulation", "hardware", "hip"])
    p.add_argument("-gpus", type=int, default=0, help="ranks to launch (default: WORLD_SIZE, else all visible GPUs)")
    for name in ("compare-to-reference", "generate-input", "print-result", "single-device"):
        p.add_argument("-" + name, dest=name.replace("-", "_"), action="store_true")
    p.add_argument("-input-directory", dest="input_directory", default=None)
    p.add_argument("-halo", type=int, default=0)
    p.add_argument("-repetitions", type=int, default=1)
    p.add_argument("-log-level", dest="log_level", type=int, choices=[0, 1, 2, 3], default=1)
    p.add_argument("-options", default=None, help="backend tuning overrides, e.g. 'fuse=2'")
    p.add_argument("-reference-checker", dest="reference_checker", default=os.environ.get("SF_REFERENCE_CHECKER"),
                   help="module:function(stencil_file, input_arrays) -> {output: ndarray}")
    return p


def visible_gpus():
    """GPUs of this node WITHOUT touching the HIP runtime (the launching process must not
    initialise a GPU: its ranks are started as fresh processes): the KFD topology lists
    every node, GPUs are the ones with SIMDs; *_VISIBLE_DEVICES narrow the count."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        value = os.environ.get(var)
        if value is not None:
            return len([v for v in value.split(",") if v.strip() != ""])
    count, nodes = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(nodes):
            with open(os.path.join(nodes, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
    except (OSError, ValueError):
        pass
    return count


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = vars(build_parser().parse_args(argv))
    gpus = args.pop("gpus")
    if "WORLD_SIZE" not in os.environ:
        # the launching process: no GPU is touched here, the ranks are fresh processes
        from stencilflow_amd.run_distributed_program import launch
        if gpus <= 0:
            gpus = visible_gpus()
            if gpus <= 0:
                raise SystemExit("no GPU found in the KFD topology: pass -gpus N")
        return launch(argv, gpus)
    import stencilflow_amd
    from stencilflow_amd.run_distributed_program import run_distributed_program
    from stencilflow_amd.run_program import load_reference_backend
    if gpus and gpus != int(os.environ["WORLD_SIZE"]):
        raise SystemExit("-gpus {} does not match WORLD_SIZE {}".format(gpus, os.environ["WORLD_SIZE"]))
    checker = args.pop("reference_checker")
    if checker:
        load_reference_backend(checker)
    args["log_level"] = stencilflow_amd.LogLevel(args["log_level"])
    result = run_distributed_program(**args)
    return 0 if result in (0, None) else 1


if __name__ == "__main__":
    sys.exit(main())
