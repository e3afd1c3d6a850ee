#!/usr/bin/env python3
"""Analytic report of a stencil program for the GPU backend: operation counts,
minimum off-chip volume (as the reference's bin/report.py:15-57 prints them)
and, instead of FPGA cycles, the HBM-roofline lower bound and the launch
schedule the planner chooses."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("stencil_file")
    p.add_argument("-options", default=None)
    p.add_argument("-hbm-tbps", type=float, default=8.0)
    a = p.parse_args()
    chain = sf.KernelChainGraph(a.stencil_file)
    chain.report()
    print("  runtime lower bound at {} TB/s: {:.6f} s".format(
        a.hbm_tbps, chain.runtime_lower_bound(a.hbm_tbps * 1e12)))
    with Plan(lower(chain), options=a.options) as plan:
        print(plan.describe())
