#!/usr/bin/env python3
"""Generates tests/golden/simulator_unit.json: the six float32 programs whose Simulator vectors on SIGNED random
data had to be compared under a relaxed rule (tests/test_reference_vectors.py: CANCELLING -- sums that cancel to
~0 cannot agree to 1e-6 of themselves across two roundings of the sum), run again by the REFERENCE's own
``stencilflow.simulator.Simulator`` on data drawn from [0, 1): nothing cancels, so the per-point rule of
BASELINE.json's north_star (1e-6 relative to the point's own value) applies to them as it stands
("the fix is data, not tolerance").

The programs are the ones authored in make_simulator_fixtures.py, make_simulator_chain_fixtures.py and
make_simulator_wide_fixtures.py, unchanged except for the values of their array inputs.  Runs only in the build
container (imports /root/reference under the shims of make_reference_fixtures.py); the fixture holds data only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_simulator_chain_fixtures as chains  # noqa: E402
import make_simulator_fixtures as base  # noqa: E402
import make_simulator_wide_fixtures as wide  # noqa: E402
from make_reference_fixtures import REFERENCE, install_shims  # noqa: E402

SEED = 20261004
CANCELLING = ("f32_chain2", "f32_chain8", "f32_jacobi7", "f32_weighted_bc", "f32_hotspot2", "f32_wide_cross2")


def unit_twin(prog, rng):
    twin = json.loads(json.dumps(prog))
    dims = twin["dimensions"]
    for name, desc in twin["inputs"].items():
        if isinstance(desc["data"], list):
            shape = [dims[d] for d in range(len(dims))] if "input_dims" not in desc else None
            count = len(desc["data"])
            dtype = np.float32 if desc["data_type"] == "float32" else np.float64
            desc["data"] = [float(x) for x in rng.random(count).astype(dtype)]
            assert shape is None or int(np.prod(shape)) == count
    return twin


def main():
    import tempfile
    install_shims()
    sys.path.insert(0, REFERENCE)
    authored = {}
    for mod in (base, chains, wide):
        authored.update(mod.authored_programs())
    rng = np.random.default_rng(SEED)
    vectors = {"source": "reference stencilflow.simulator.Simulator (kernel.py:700-709), array inputs in [0, 1)",
               "numpy": np.__version__, "seed": SEED, "programs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name in CANCELLING:
            twin = unit_twin(authored[name], rng)
            result, cycles = base.run_simulator(name + "_unit", twin, tmp)
            if result is None:
                raise SystemExit("{}: the Simulator did not finish".format(name))
            vectors["programs"][name + "_unit"] = {"program": twin, "cycles": cycles, "result": result}
            print("{}_unit: {} cycles, outputs {}".format(name, cycles, sorted(result)))
    with open(os.path.join(HERE, "simulator_unit.json"), "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote simulator_unit.json with", len(vectors["programs"]), "programs")


if __name__ == "__main__":
    main()
