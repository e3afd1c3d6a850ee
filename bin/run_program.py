#!/usr/bin/env python3
"""Command line of the driver; same flags as the reference's bin/run_program.py
(:12-37).  FPGA-only flags are accepted and ignored."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import stencilflow_amd  # noqa: E402
from stencilflow_amd.run_program import run_program  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("stencil_file")
    parser.add_argument("mode", choices=["emulation", "hardware", "hip"])
    parser.add_argument("-run-simulation", action="store_true")
    parser.add_argument("-compare-to-reference", action="store_true")
    parser.add_argument("-input-directory")
    parser.add_argument("-use-cached-sdfg", dest="use_cached_sdfg",
                        action="store_true")
    parser.add_argument("-skip-execution", dest="skip_execution",
                        action="store_true")
    parser.add_argument("-generate-input", action="store_true")
    parser.add_argument("-halo", type=int, default=0)
    parser.add_argument("-repetitions", type=int, default=1)
    parser.add_argument("-synthetic-reads", type=float, default=None)
    parser.add_argument("-specialize-scalars", dest="specialize_scalars",
                        action="store_true")
    parser.add_argument("-plot", action="store_true")
    parser.add_argument("-log-level", type=int, choices=[0, 1, 2, 3],
                        default=1)
    parser.add_argument("-print-result", dest="print_result",
                        action="store_true")
    parser.add_argument("-xilinx", dest="xilinx", action="store_true")
    parser.add_argument("-device", type=int, default=0)
    parser.add_argument("-options", type=str, default=None,
                        help="backend tuning overrides, e.g. 'fuse=2;k1.rj=5'")
    args = parser.parse_args()
    args.log_level = stencilflow_amd.LogLevel(args.log_level)
    if args.compare_to_reference:
        # the CPU checker lives with the tests, not in the product
        from tests.reference_provider import register
        register()
    sys.exit(run_program(**vars(args)))
