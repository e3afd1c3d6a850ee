#!/usr/bin/env python3
"""Generates tests/golden/simulator_chains.json: outputs of the REFERENCE's own
``stencilflow.simulator.Simulator`` (imported from /root/reference under the shims of
make_reference_fixtures.py) on two chains authored here in the reference's format:
  f64_chain3    float64 diffusion -> advection -> laplacian, the structure of BASELINE.json's
                configs[3] (C5), on a 5 x 6 x 6 grid with literal coefficients;
  f32_hotspot2  two float32 operators of the workload generator's hotspot shape
                (bin/synthesize.py:133-165), both reading the same auxiliary field at the centre.
The fixture holds data only: the programs and the numbers the Simulator returned.
Runs only in the build container; tests/test_reference_vectors.py reads the committed JSON."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_reference_fixtures import REFERENCE, install_shims  # noqa: E402
import make_simulator_fixtures as base  # noqa: E402


def authored_programs():
    rng = np.random.default_rng(20261004)

    def field(dims, dtype):
        return [float(x) for x in np.asarray(rng.uniform(-1, 1, size=dims), dtype=dtype).ravel()]

    def bc(*names):
        return {n: {"type": "constant", "value": 0.0} for n in names}
    dims = [5, 6, 6]
    progs = {"f64_chain3": {
        "inputs": {"a": {"data": field(dims, np.float64), "data_type": "float64"}}, "outputs": ["lap"], "dimensions": dims,
        "program": {
            "diff": {"computation_string": "diff = 0.4*a[i,j,k] + 0.1*a[i-1,j,k] + 0.1*a[i+1,j,k] + 0.1*a[i,j-1,k] + "
                                           "0.1*a[i,j+1,k] + 0.1*a[i,j,k-1] + 0.1*a[i,j,k+1]",
                     "boundary_conditions": bc("a"), "data_type": "float64"},
            "adv": {"computation_string": "adv = diff[i,j,k] - 0.3*(diff[i,j,k]-diff[i-1,j,k]) - "
                                          "0.2*(diff[i,j,k]-diff[i,j-1,k]) - 0.1*(diff[i,j,k]-diff[i,j,k-1])",
                    "boundary_conditions": bc("diff"), "data_type": "float64"},
            "lap": {"computation_string": "lap = adv[i-1,j,k]+adv[i+1,j,k]+adv[i,j-1,k]+adv[i,j+1,k]+adv[i,j,k-1]+"
                                          "adv[i,j,k+1]-6.0*adv[i,j,k]",
                    "boundary_conditions": bc("adv"), "data_type": "float64"}}}}

    def hot(o, s):
        return ("{o} = {s}[i,j,k] + 0.125 * (p[i,j,k] + ({s}[i,j+1,k] + {s}[i,j-1,k] - 2.0 * {s}[i,j,k]) * 0.3 + "
                "({s}[i,j,k+1] + {s}[i,j,k-1] - 2.0 * {s}[i,j,k]) * 0.2 + ({s}[i+1,j,k] + {s}[i-1,j,k] - 2.0 * {s}[i,j,k]) * 0.1 + "
                "(0.5 - {s}[i,j,k]) * 0.05)").format(o=o, s=s)
    dims = [5, 6, 8]
    progs["f32_hotspot2"] = {
        "inputs": {"a": {"data": field(dims, np.float32), "data_type": "float32"},
                   "p": {"data": field(dims, np.float32), "data_type": "float32"}},
        "outputs": ["b1"], "dimensions": dims,
        "program": {"b0": {"computation_string": hot("b0", "a"), "boundary_conditions": bc("a", "p"), "data_type": "float32"},
                    "b1": {"computation_string": hot("b1", "b0"), "boundary_conditions": bc("b0", "p"), "data_type": "float32"}}}
    return progs


def main():
    install_shims()
    sys.path.insert(0, REFERENCE)
    out = {"source": "reference stencilflow.simulator.Simulator (kernel.py:700-709)", "numpy": np.__version__, "programs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog in authored_programs().items():
            result, cycles = base.run_simulator(name, prog, tmp, max_cycles=60000)
            if result is None:
                raise SystemExit(name + ": the simulator did not finish")
            out["programs"][name] = {"program": prog, "cycles": cycles, "result": result}
            print("{}: {} cycles".format(name, cycles))
    with open(os.path.join(HERE, "simulator_chains.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
