"""How far apart are the two typings the reference's tasklet text admits?

The reference builds ``v = <bc> if <oob> else v_in`` per access
(stencilflow/stencil/cpu.py:89-102) and leaves the types to DaCe's C++: a
boundary literal ``0.0`` makes every neighbour a ``double`` and the six-term sum
a double sum rounded once (the contract of the oracle and of the
HIP kernels); a literal ``0`` (what bin/synthesize.py:219 writes) keeps the sum
in ``float``, rounded after every add.  Both programs define the same
mathematical operator; this module applies both to the same data and reports
``max |d - f| / max(|d|, |f|)`` after a number of operators -- the *rounding
envelope*: a result within 1e-6 of one typing is within 1e-6 of the reference
whatever DaCe does iff the envelope stays below 1e-6.

Test infrastructure (uses the C oracle).  As a script it prints the table kept in
profiles/r02_rounding_envelope.log:

    python -m tests.rounding_envelope 512 1000
"""
import sys

import numpy as np

from oracle import c_oracle, numpy_oracle as npo
from stencilflow_amd import programs

SEED = 20261003
BLOCK = 8


def initial(shape, data):
    if data == "ones":  # the reference's own input ("constant:1.0")
        return np.ones(shape, np.float32)
    return np.random.default_rng(SEED).random(shape, dtype=np.float32)


def envelope(shape, stages, data, report_at=()):
    """Returns {stage: max relative difference} at the stages of `report_at`
    (multiples of 8) and at `stages`."""
    assert stages % BLOCK == 0
    double_sum = c_oracle.CompiledReference(programs.jacobi3d(shape, BLOCK, bc_value=0.0))
    float_sum = c_oracle.CompiledReference(programs.jacobi3d(shape, BLOCK, bc_value=0))
    out = "b{}".format(BLOCK - 1)
    d = f = initial(shape, data)
    table = {}
    for s in range(BLOCK, stages + 1, BLOCK):
        d = double_sum.run({"a": d})[out]
        f = float_sum.run({"a": f})[out]
        if s in report_at or s == stages:
            table[s] = npo.max_rel_err(d, f)
    return table


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    stages = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    marks = [s for s in (8, 16, 32, 64, 104, 200, 304, 400, 600, 800, 1000) if s <= stages]
    print("# jacobi3d {0}x{0}x{0} float32, coefficient 0.16666666, BC 0: double-sum typing "
          "(literal 0.0) against float-sum typing (literal 0), C oracle, strict IEEE".format(n))
    for data in ("ones", "random"):
        table = envelope((n, n, n), stages, data, marks)
        crossed = next((s for s in sorted(table) if table[s] > 1e-6), None)
        for s in sorted(table):
            print("{:>7} data  {:5d} operators  max rel diff {:.3e}".format(data, s, table[s]), flush=True)
        print("{:>7} data  first reported depth above 1e-6: {}".format(data, crossed))


if __name__ == "__main__":
    main()
