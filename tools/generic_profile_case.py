#!/usr/bin/env python3
"""GPU: one operator per launch on the generic kernel (jacobi3d 512^3 f32, option generic_only=1) --
the command tools/profile.sh wraps when the generic kernel alone is to be profiled.  Prints ms per operator."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
with tempfile.TemporaryDirectory() as tmp:
    path = programs.write_program(programs.jacobi3d((512,512,512), 8), os.path.join(tmp, "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
plan = Plan(sfir, options="generic_only=1")
x = np.random.default_rng(0).random((512,512,512), dtype=np.float32)
plan.upload([x])
for _ in range(3):
    plan.execute(1); plan.synchronize()
print("ms per op", plan.elapsed_ms()/8)
