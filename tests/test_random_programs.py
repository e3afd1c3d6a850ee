"""Randomly generated programs (tests/random_programs.py): the two oracle
implementations agree bit for bit (CPU), the front end lowers every program and
the library compiles it (CPU), and the HIP backend agrees with the oracle bit
for bit (GPU)."""
import json
import os

import numpy as np
import pytest

import stencilflow_amd as sf
from oracle import c_oracle, numpy_oracle as npo
from stencilflow_amd import programs
from stencilflow_amd.backend import CompiledProgram, Plan
from stencilflow_amd.lowering import lower
from tests.random_programs import random_inputs, random_program

CPU_SEEDS = list(range(100, 124))
GPU_SEEDS = list(range(100, 160))


@pytest.mark.parametrize("seed", CPU_SEEDS)
def test_oracles_agree_on_random_programs(seed, tmp_path):
    prog = random_program(seed)
    ins = random_inputs(prog, seed)
    a = npo.run_reference(prog, inputs=ins, return_all=True)
    b = c_oracle.CompiledReference(prog).run(inputs=ins, return_all=True)
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    # and the product front end accepts it
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    chain = sf.KernelChainGraph(path)
    order = [k.name for k in chain.topological_kernels()]
    assert sorted(order) == sorted(prog["program"])
    for pos, kname in enumerate(order):  # every producer precedes its consumers
        for dep in npo._reads(prog, kname):
            if dep in prog["program"]:
                assert order.index(dep) < pos
    with Plan(lower(chain)) as plan:
        assert sorted(plan.output_names) == sorted(prog["outputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GPU_SEEDS)
def test_hip_matches_oracle_on_random_programs(seed, tmp_path):
    prog = random_program(seed)
    ins = random_inputs(prog, seed)
    want = npo.run_reference(prog, inputs=ins)
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    chain = sf.KernelChainGraph(path)
    compiled = CompiledProgram(chain)
    outs = {n: np.zeros(prog["dimensions"], dtype=chain.program[n]["data_type"].type)
            for n in chain.outputs}
    kwargs = {}
    for name, desc in chain.inputs.items():
        if len(desc["input_dims"]) == 0:
            kwargs[name] = ins[name]
        else:
            kwargs[name + "_host"] = np.ascontiguousarray(ins[name])
    kwargs.update({n + "_host": a for n, a in outs.items()})
    compiled(**kwargs)
    compiled.close()
    for k in want:
        assert np.array_equal(outs[k], want[k], equal_nan=True), (
            seed, k, npo.max_rel_err(want[k], outs[k]), json.dumps(prog)[:400])
