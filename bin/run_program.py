#!/usr/bin/env python3
"""Command line of the driver.  Flag names, defaults and the positional
arguments are those of the reference's bin/run_program.py (:12-37) so existing
invocations keep working; flags that only steer the FPGA toolchain are accepted
and ignored.  Extra: ``-device``, ``-options`` (backend tuning overrides) and
``-reference-checker module:function`` (or ``$SF_REFERENCE_CHECKER``): the CPU
checker ``-compare-to-reference`` compares against.  The product ships none --
results never come from a CPU path -- so the flag needs one named explicitly,
e.g. ``-reference-checker tests.reference_provider:reference_outputs`` in a
checkout of this repository."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import stencilflow_amd  # noqa: E402

# (flag, kwargs) -- single-dash long options, as in the reference
SWITCHES = ["run-simulation", "compare-to-reference", "use-cached-sdfg", "skip-execution",
            "generate-input", "specialize-scalars", "plot", "print-result", "xilinx"]
VALUED = [
    ("input-directory", dict(default=None)),
    ("halo", dict(type=int, default=0)),
    ("repetitions", dict(type=int, default=1)),
    ("synthetic-reads", dict(type=float, default=None)),
    ("log-level", dict(type=int, choices=[0, 1, 2, 3], default=1)),
    ("device", dict(type=int, default=0)),
    ("options", dict(type=str, default=None, help="e.g. 'fuse=2;k1.rj=5'")),
    ("reference-checker", dict(type=str, default=os.environ.get("SF_REFERENCE_CHECKER"),
                               help="module:function(stencil_file, input_arrays) -> {output: ndarray}")),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__)
    parser.add_argument("stencil_file")
    parser.add_argument("mode", choices=["emulation", "hardware", "hip"])
    for name in SWITCHES:
        parser.add_argument("-" + name, dest=name.replace("-", "_"), action="store_true")
    for name, kw in VALUED:
        parser.add_argument("-" + name, dest=name.replace("-", "_"), **kw)
    return parser


def main(argv=None):
    args = vars(build_parser().parse_args(argv))
    args["log_level"] = stencilflow_amd.LogLevel(args["log_level"])
    checker = args.pop("reference_checker")
    if checker:
        from stencilflow_amd.run_program import load_reference_backend
        load_reference_backend(checker)
    return stencilflow_amd.run_program(**args)


if __name__ == "__main__":
    sys.exit(main())
