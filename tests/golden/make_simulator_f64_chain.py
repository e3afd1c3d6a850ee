#!/usr/bin/env python3
"""Generates tests/golden/simulator_f64_chain.json: the output of the REFERENCE's own
``stencilflow.simulator.Simulator`` (imported from /root/reference under the shims of
make_reference_fixtures.py) on a float64 diffusion -> advection -> laplacian chain -- the
structure of BASELINE.json's configs[3] (C5) on a 5 x 6 x 6 grid with literal coefficients.
The fixture holds data only: the program authored here and the numbers the Simulator returned.
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


def authored_program():
    rng = np.random.default_rng(20261004)
    dims = [5, 6, 6]
    field = [float(x) for x in rng.uniform(-1, 1, size=dims).ravel()]

    def bc(name):
        return {name: {"type": "constant", "value": 0.0}}
    return {
        "inputs": {"a": {"data": field, "data_type": "float64"}}, "outputs": ["lap"], "dimensions": dims,
        "program": {
            "diff": {"computation_string": "diff = 0.4*a[i,j,k] + 0.1*a[i-1,j,k] + 0.1*a[i+1,j,k] + 0.1*a[i,j-1,k] + "
                                           "0.1*a[i,j+1,k] + 0.1*a[i,j,k-1] + 0.1*a[i,j,k+1]",
                     "boundary_conditions": bc("a"), "data_type": "float64"},
            "adv": {"computation_string": "adv = diff[i,j,k] - 0.3*(diff[i,j,k]-diff[i-1,j,k]) - "
                                          "0.2*(diff[i,j,k]-diff[i,j-1,k]) - 0.1*(diff[i,j,k]-diff[i,j,k-1])",
                    "boundary_conditions": bc("diff"), "data_type": "float64"},
            "lap": {"computation_string": "lap = adv[i-1,j,k]+adv[i+1,j,k]+adv[i,j-1,k]+adv[i,j+1,k]+adv[i,j,k-1]+"
                                          "adv[i,j,k+1]-6.0*adv[i,j,k]",
                    "boundary_conditions": bc("adv"), "data_type": "float64"}}}


def main():
    install_shims()
    sys.path.insert(0, REFERENCE)
    prog = authored_program()
    with tempfile.TemporaryDirectory() as tmp:
        result, cycles = base.run_simulator("f64_chain3", prog, tmp, max_cycles=40000)
    if result is None:
        raise SystemExit("the simulator did not finish")
    out = {"source": "reference stencilflow.simulator.Simulator (kernel.py:700-709)", "numpy": np.__version__,
           "program": prog, "cycles": cycles, "result": result}
    with open(os.path.join(HERE, "simulator_f64_chain.json"), "w") as f:
        json.dump(out, f)
    print("f64_chain3: {} cycles".format(cycles))


if __name__ == "__main__":
    main()
