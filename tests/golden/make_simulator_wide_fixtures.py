#!/usr/bin/env python3
"""Generates tests/golden/simulator_wide.json: outputs of the REFERENCE's own
``stencilflow.simulator.Simulator`` (imported from /root/reference under the shims of
make_reference_fixtures.py) on radius-2 star programs -- the operators the reference's
generator emits for an extent of 2 (bin/synthesize.py:19-31,91-104) -- authored here in the
reference's format:
  f32_wide_cross_exact   the generator's 13-point cross text (float sum, coefficient literal) on data
                         on which float32 arithmetic is exact, non-zero boundary constant;
  f32_wide_cross2        two such operators chained on random float32 data (what a fused
                         wide-star launch evaluates);
  f64_wide_diffusion     the generator's diffusion text with literal coefficients, float64.
(No 2-D program: the Simulator never terminates on 2-D programs -- reference
test/test_stencilflow.py:202 says as much; tried with the 2-D cross of radius 2, 60 000 cycles.)
The fixture holds data only: the programs and the numbers the Simulator returned.  Runs only
in the build container; tests/test_reference_vectors.py reads the committed JSON (round 3: the
vectors that pin kernels/wstar3d.h against the reference itself)."""
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
    rng = np.random.default_rng(20261005)

    def field(dims, dtype, exact=False):
        v = rng.integers(-8, 9, size=dims).astype(np.float64) / 4.0 if exact else rng.uniform(-1, 1, size=dims)
        return [float(x) for x in np.asarray(v, dtype=dtype).ravel()]

    def bc(name, value):
        return {name: {"type": "constant", "value": value}}

    def cross(o, s, coef):
        return ("{o} = {c}*({s}[i-2, j, k] + {s}[i-1, j, k] + {s}[i+1, j, k] + {s}[i+2, j, k] + {s}[i, j-2, k] + "
                "{s}[i, j-1, k] + {s}[i, j+1, k] + {s}[i, j+2, k] + {s}[i, j, k-2] + {s}[i, j, k-1] + {s}[i, j, k+1] + "
                "{s}[i, j, k+2])").format(o=o, s=s, c=coef)
    dims = [6, 7, 8]
    progs = {
        "f32_wide_cross_exact": {
            "inputs": {"a": {"data": field(dims, np.float32, True), "data_type": "float32"}}, "outputs": ["b0"],
            "dimensions": dims,
            "program": {"b0": {"computation_string": cross("b0", "a", "0.125"), "boundary_conditions": bc("a", 1.5),
                               "data_type": "float32"}}},
        "f32_wide_cross2": {
            "inputs": {"a": {"data": field(dims, np.float32), "data_type": "float32"}}, "outputs": ["b1"],
            "dimensions": dims,
            "program": {"b0": {"computation_string": cross("b0", "a", "0.08333333333333333"), "boundary_conditions": bc("a", 0),
                               "data_type": "float32"},
                        "b1": {"computation_string": cross("b1", "b0", "0.08333333333333333"),
                               "boundary_conditions": bc("b0", 0), "data_type": "float32"}}},
        "f64_wide_diffusion": {
            "inputs": {"a": {"data": field(dims, np.float64), "data_type": "float64"}}, "outputs": ["d"],
            "dimensions": dims,
            "program": {"d": {"computation_string":
                              "d = 0.4*a[i, j, k] + 0.02*a[i-2, j, k] + 0.08*a[i-1, j, k] + 0.08*a[i+1, j, k] + 0.02*a[i+2, j, k] + "
                              "0.03*a[i, j-2, k] + 0.07*a[i, j-1, k] + 0.07*a[i, j+1, k] + 0.03*a[i, j+2, k] + 0.01*a[i, j, k-2] + "
                              "0.09*a[i, j, k-1] + 0.09*a[i, j, k+1] + 0.01*a[i, j, k+2]",
                              "boundary_conditions": bc("a", 0.25), "data_type": "float64"}}},
    }
    return progs


def main():
    install_shims()
    sys.path.insert(0, REFERENCE)
    out = {"source": "reference stencilflow.simulator.Simulator (kernel.py:700-709)", "numpy": np.__version__, "programs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog in authored_programs().items():
            result, cycles = base.run_simulator(name, prog, tmp, max_cycles=60000)
            if result is None:
                print("{}: the simulator did not finish within {} cycles -- left out".format(name, cycles))
                continue
            out["programs"][name] = {"program": prog, "cycles": cycles, "result": result}
            print("{}: {} cycles".format(name, cycles))
    with open(os.path.join(HERE, "simulator_wide.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
