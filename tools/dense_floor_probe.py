#!/usr/bin/env python3
"""GPU: what the dense kernel's skeleton costs (planes through registers into the LDS ring, one barrier per
plane, results out) apart from the operator's arithmetic: a 5-term and a 27-term operator over offsets at
distance 2, 512^3 float32, in the kernel's two forms and on the generic kernel (profiles/r03_dense_floor.log:
0.20-0.22 ms = the HBM bound of one read and one write; the 125-point box takes 0.595 ms).
usage (repository root): python tools/dense_floor_probe.py"""
import sys, os, tempfile, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
n = 512
offs_small = [(-2,-2,-2),(2,2,2),(0,0,0),(1,-2,0),(-1,2,1)]
offs_mid = [(i,j,k) for i in (-2,0,2) for j in (-2,0,2) for k in (-2,0,2)]
for label, offs in (("5 terms", offs_small), ("27 terms (stride 2)", offs_mid)):
    terms = ["a[%s]" % ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip("ijk", off)) for off in offs]
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": "float32"}}, "outputs": ["b0"], "dimensions": [n, n, n],
            "program": {"b0": {"computation_string": "b0 = 0.1 * (" + " + ".join(terms) + ")", "boundary_conditions": {"a": {"type": "constant", "value": 0}}, "data_type": "float32"}}}
    with tempfile.TemporaryDirectory() as tmp:
        sfir = lower(sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json"))))
    x = np.random.default_rng(1).uniform(-1, 1, (n, n, n)).astype(np.float32)
    for opt in ({}, {"dense.sum": 0}, {"dense": 0}):
        with Plan(sfir, options=opt) as plan:
            plan.upload([x]); plan.execute(2); plan.synchronize()
            t0 = time.perf_counter(); plan.execute(10); plan.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(json.dumps({"case": label, "opt": opt, "ms": round(ms, 4), "launch": plan.describe().splitlines()[1].strip()[:150]}), flush=True)
