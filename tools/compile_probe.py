#!/usr/bin/env python3
"""Which hipRTC compiles the kernels, and is its output stable?  Prints the
register count of the C3 kernel and the hiprtc / comgr libraries mapped into the
process; `--torch-first` imports torch before the library is loaded (as
bench.py does)."""
import os
import sys
import tempfile

if "--torch-first" in sys.argv:
    import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    path = programs.write_program(programs.jacobi3d((512, 512, 512), 4), os.path.join(tmp, "p.json"))
    plan = Plan(lower(sf.KernelChainGraph(path)))
libs = sorted({line.split()[-1] for line in open("/proc/self/maps")
               if any(k in line for k in ("hiprtc", "comgr", "amdhip64"))})
print(plan.describe().splitlines()[1].strip()[-58:], "|", " ".join(libs))
