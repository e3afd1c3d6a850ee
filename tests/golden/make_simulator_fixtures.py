#!/usr/bin/env python3
"""Generates tests/golden/simulator_vectors.json: output vectors of the
REFERENCE's own ``stencilflow.simulator.Simulator`` (imported from
/root/reference under the shims of make_reference_fixtures.py) on small
float32 / mixed-dtype 3-D programs authored here in the reference's JSON
format.

Why: the reference stores no expected outputs (its tests are run-time
differential), and its CPU path needs DaCe, which is absent.  The Simulator is
the one numeric evaluator of the reference that runs in this container.  It
evaluates ``self.data_type(calculator.eval_expr(var_map, computation))``
(stencilflow/kernel.py:700-709): NumPy scalar arithmetic on the values of the
input arrays, with Python-float literals -- so its intermediate precision is
NumPy's, not DaCe's C++ (under NumPy >= 2, NEP 50: float32 op python-float ->
float32).  The vectors therefore pin indexing, boundary selection, operand
order and dtype casts of float32 / mixed programs exactly, and the arithmetic
to within the 1e-6 relative tolerance of BASELINE.json's north_star; programs
whose arithmetic is exact in float32 (`*_exact`) pin it bit for bit.

Runs only in the build container; the tests read the committed JSON.  The
fixture holds data only: the programs authored below and the numbers the
reference's Simulator returned for them.
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_reference_fixtures import REFERENCE, install_shims  # noqa: E402

SEED = 20261003


def _field(rng, shape, dtype, exact):
    if exact:
        # small dyadic rationals: every sum/product below is exact in float32
        v = rng.integers(-8, 9, size=shape).astype(np.float64) / 4.0
    else:
        v = rng.uniform(-1.0, 1.0, size=shape)
    return [float(x) for x in np.asarray(v, dtype=dtype).ravel()]


def _bc(fields, value=0.0, kind="constant"):
    return {f: {"type": kind, "value": value} for f in fields}


def authored_programs():
    rng = np.random.default_rng(SEED)
    progs = {}
    dims = [4, 5, 6]
    jac = ("{o} = 0.16666666 * ({s}[i-1,j,k] + {s}[i+1,j,k] + {s}[i,j-1,k] + "
           "{s}[i,j+1,k] + {s}[i,j,k-1] + {s}[i,j,k+1])")
    # 1. the benchmark operator itself, float32, random data, BC 0.0
    progs["f32_jacobi7"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"}},
        "outputs": ["b"], "dimensions": dims,
        "program": {"b": {"computation_string": jac.format(o="b", s="a"),
                          "boundary_conditions": _bc(["a"]), "data_type": "float32"}},
    }
    # 2. the same with data on which float32 arithmetic is exact (0.25 coefficient)
    progs["f32_jacobi7_exact"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, True), "data_type": "float32"}},
        "outputs": ["b"], "dimensions": dims,
        "program": {"b": {"computation_string":
                          "b = 0.25 * (a[i-1,j,k] + a[i+1,j,k] + a[i,j-1,k] + a[i,j+1,k] + a[i,j,k-1] + a[i,j,k+1])",
                          "boundary_conditions": _bc(["a"], 1.5), "data_type": "float32"}},
    }
    # 3. weighted 7-point with centre and a non-zero boundary constant
    progs["f32_weighted_bc"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"}},
        "outputs": ["w"], "dimensions": dims,
        "program": {"w": {"computation_string":
                          "w = 0.5 * a[i,j,k] + 0.125 * (a[i-1,j,k] + a[i+1,j,k]) - 0.0625 * (a[i,j-1,k] + a[i,j+1,k]) "
                          "+ 0.03125 * a[i,j,k-1] + 0.03125 * a[i,j,k+1]",
                          "boundary_conditions": _bc(["a"], 2.75), "data_type": "float32"}},
    }
    # 4. two-operator float32 chain (the benchmark's structure at depth 2)
    progs["f32_chain2"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"}},
        "outputs": ["b1"], "dimensions": dims,
        "program": {
            "b0": {"computation_string": jac.format(o="b0", s="a"),
                   "boundary_conditions": _bc(["a"]), "data_type": "float32"},
            "b1": {"computation_string": jac.format(o="b1", s="b0"),
                   "boundary_conditions": _bc(["b0"]), "data_type": "float32"},
        },
    }
    # 4b. the benchmark's chain at BASELINE configs[0]'s depth (8 operators)
    dims8 = [6, 6, 8]
    progs["f32_chain8"] = {
        "inputs": {"a": {"data": _field(rng, dims8, np.float32, False), "data_type": "float32"}},
        "outputs": ["b7"], "dimensions": dims8,
        "program": {
            "b{}".format(t): {"computation_string": jac.format(o="b{}".format(t), s="a" if t == 0 else "b{}".format(t - 1)),
                              "boundary_conditions": _bc(["a" if t == 0 else "b{}".format(t - 1)]),
                              "data_type": "float32"}
            for t in range(8)
        },
    }
    # 5. mixed dtypes: float32 and float64 fields into a float64 result ...
    progs["mixed_to_f64"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"},
                   "c": {"data": _field(rng, dims, np.float64, False), "data_type": "float64"}},
        "outputs": ["hi"], "dimensions": dims,
        "program": {
            "hi": {"computation_string": "hi = 0.5 * a[i,j,k] + c[i,j,k+1] - c[i-1,j,k]",
                   "boundary_conditions": _bc(["a", "c"], 0.0), "data_type": "float64"},
        },
    }
    # ... and into a float32 result (the cast on the store is the point).
    # (`a` is read at the centre too: with `a[i,j-1,k]` as its only access the
    # Simulator's FPGA buffer model hands the kernel the element one position
    # back in flat order -- a[0,0,5] for point (0,1,0) -- instead of a[i,j-1,k];
    # that is the buffer model's, not cpu.py's, and such a program is left out)
    progs["mixed_to_f32"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"},
                   "c": {"data": _field(rng, dims, np.float64, False), "data_type": "float64"}},
        "outputs": ["lo"], "dimensions": dims,
        "program": {
            "lo": {"computation_string": "lo = c[i,j,k] * 0.75 - a[i,j-1,k] + c[i+1,j,k] + 0.5 * a[i,j,k]",
                   "boundary_conditions": _bc(["a", "c"], 1.0), "data_type": "float32"},
        },
    }
    # 6. 27-point box neighbourhood subset (corners and edges), float32, exact data
    progs["f32_box_exact"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, True), "data_type": "float32"}},
        "outputs": ["x"], "dimensions": dims,
        "program": {"x": {"computation_string":
                          "x = a[i-1,j-1,k-1] + a[i+1,j+1,k+1] + a[i,j-1,k+1] + a[i-1,j,k+1] + a[i+1,j-1,k] + 2.0 * a[i,j,k]",
                          "boundary_conditions": _bc(["a"], -0.5), "data_type": "float32"}},
    }
    # 7. fan-out / fan-in with a ternary on data
    progs["f32_fork_join"] = {
        "inputs": {"a": {"data": _field(rng, dims, np.float32, False), "data_type": "float32"}},
        "outputs": ["z"], "dimensions": dims,
        "program": {
            "p": {"computation_string": "p = a[i,j,k] + a[i,j,k+1]",
                  "boundary_conditions": _bc(["a"], 0.0), "data_type": "float32"},
            "q": {"computation_string": "q = a[i,j,k] - a[i+1,j,k]",
                  "boundary_conditions": _bc(["a"], 0.0), "data_type": "float32"},
            "z": {"computation_string": "z = p[i,j,k] if p[i,j,k] > q[i,j,k] else q[i,j-1,k]",
                  "boundary_conditions": _bc(["p", "q"], 0.25), "data_type": "float32"},
        },
    }
    return progs


def run_simulator(name, prog, tmpdir, max_cycles=20000):
    import stencilflow
    from stencilflow.kernel_chain_graph import KernelChainGraph
    from stencilflow.simulator import Simulator
    from stencilflow.log_level import LogLevel
    path = os.path.join(tmpdir, name + ".json")
    with open(path, "w") as f:
        json.dump(prog, f)
    desc = stencilflow.parse_json(path)
    chain = KernelChainGraph(path)
    sim = Simulator(program_name=name, program_description=desc,
                    input_nodes=chain.input_nodes, kernel_nodes=chain.kernel_nodes,
                    output_nodes=chain.output_nodes, dimensions=chain.dimensions,
                    write_output=False, log_level=LogLevel.NO_LOG)
    # Simulator.simulate() (simulator.py:176-220) without its prints, with a cycle
    # cap: the reference's simulator deadlocks on some programs
    sim.initialize()
    cycles = 0
    while not sim.all_done():
        sim.step_execution()
        cycles += 1
        if cycles > max_cycles:
            return None, cycles
    out = {}
    for k, v in sim.get_result().items():
        arr = np.asarray(v)
        out[k] = {"dtype": str(arr.dtype), "values": [float(x) for x in arr.ravel()]}
    return out, cycles


def main():
    import tempfile
    install_shims()
    sys.path.insert(0, REFERENCE)
    vectors = {"source": "reference stencilflow.simulator.Simulator (kernel.py:700-709)",
               "numpy": np.__version__, "seed": SEED, "programs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, prog in authored_programs().items():
            result, cycles = run_simulator(name, prog, tmp)
            if result is None:
                print("{}: simulator did not finish within {} cycles -- skipped".format(name, cycles))
                continue
            vectors["programs"][name] = {"program": prog, "cycles": cycles, "result": result}
            print("{}: {} cycles, outputs {}".format(name, cycles, sorted(result)))
        # BASELINE.json configs[0] itself: jacobi3d 32^3, 8 operators, float32 -- the
        # reference's own program file (vectorization set to 1: the Simulator models
        # W lanes per cycle, results do not depend on W, sdfg_generator.py:594-595),
        # once with the file's own input (a == 1.0) and once on random data.  About a
        # minute each; outputs are kept as raw float32 files like the reference's
        # results/<name>/<out>.dat (helper.py:249-258).
        with open(os.path.join(HERE, "programs", "jacobi3d_32x32x32_8itr_8vec.json")) as f:
            c1 = json.load(f)
        c1["vectorization"] = 1
        rnd = np.random.default_rng(SEED).random((32, 32, 32), dtype=np.float32)
        rnd.tofile(os.path.join(HERE, "c1_random_input_a.dat"))
        assert c1["inputs"]["a"]["data"] == "constant:1.0"
        for tag, data in (("c1_file_input", None), ("c1_random_input", rnd)):
            prog = json.loads(json.dumps(c1))
            # (the file's "constant:1.0" is handed over as the list of values it stands
            # for: with a "constant:" input the Simulator never terminates, its input
            # node keeps a scalar where a queue of values is expected)
            values = np.ones((32, 32, 32), np.float32) if data is None else data
            prog["inputs"]["a"]["data"] = [float(x) for x in values.ravel()]
            result, cycles = run_simulator(tag, prog, tmp, max_cycles=100000)
            if result is None:
                print("{}: simulator did not finish -- skipped".format(tag))
                continue
            out = np.array(result["b7"]["values"], dtype=result["b7"]["dtype"])
            assert out.dtype == np.float32 and out.size == 32**3
            out.tofile(os.path.join(HERE, tag + "_b7.dat"))
            vectors["large"] = vectors.get("large", {})
            vectors["large"][tag] = {"program": "programs/jacobi3d_32x32x32_8itr_8vec.json (vectorization 1)",
                                     "input": None if data is None else "c1_random_input_a.dat",
                                     "output": tag + "_b7.dat", "dtype": "float32", "shape": [32, 32, 32],
                                     "cycles": cycles}
            print("{}: {} cycles".format(tag, cycles))
    with open(os.path.join(HERE, "simulator_vectors.json"), "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote simulator_vectors.json with", len(vectors["programs"]), "programs")


if __name__ == "__main__":
    main()
