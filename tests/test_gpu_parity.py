"""Parity of the HIP backend against the oracle (-m gpu; calls go through the
C ABI of libsf_hip.so).  Floating point contract: results are produced by the
same typed expression in the same order as the oracle, so they are required to
be bit-identical for + - * / programs, and within 1e-6 relative (BASELINE.json)
where device math functions are involved."""
import os

import numpy as np
import pytest

import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import CompiledProgram
from oracle import numpy_oracle as npo

pytestmark = pytest.mark.gpu

SEED = 20261003


def _write(tmp_path, prog, name="prog"):
    return programs.write_program(prog, str(tmp_path / (name + ".json")))


def _run_gpu(path, inputs, options=None):
    chain = sf.KernelChainGraph(path)
    prog = CompiledProgram(chain, options=options)
    outs = {
        name: np.zeros(chain.dimensions[3 - chain.kernel_dimensions:],
                       dtype=chain.program[name]["data_type"].type)
        for name in chain.outputs
    }
    kwargs = {}
    for name, desc in chain.inputs.items():
        val = inputs[name]
        if len(desc["input_dims"]) == 0:
            kwargs[name] = val
        else:
            kwargs[name + "_host"] = np.ascontiguousarray(val)
    for name, arr in outs.items():
        kwargs[name + "_host"] = arr
    prog(**kwargs)
    desc = prog.plan.describe()
    prog.close()
    return outs, desc


def _inputs_of(path, programs_dir=None, rng=None):
    prog = npo.load_program(path)
    vals = {}
    for name in prog["inputs"]:
        override = None
        dims = npo._input_dims(prog, name)
        if rng is not None and dims:
            shape = npo._dims_shape(prog, dims)
            override = rng.uniform(-1, 1, shape).astype(
                npo._NP[prog["inputs"][name]["data_type"]])
        vals[name] = npo.materialise_input(prog, name, programs_dir, override)
    return vals


FIXTURES = [
    "jacobi2d_128x128", "jacobi2d_128x128_8vec", "jacobi3d_32x32x32",
    "jacobi3d_32x32x32_8itr", "jacobi3d_32x32x32_8itr_4vec",
    "jacobi3d_32x32x32_8itr_8vec", "simulator", "simulator2", "simulator3",
    "simulator4", "simulator5", "simulator6", "simulator7", "simulator8",
    "simulator9", "simulator10", "simulator11", "simulator12",
    "varying_dimensionality"
]


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_test_programs(programs_dir, name):
    """Every program of the reference's own test-suite
    (test/test_stencilflow.py:188-224), inputs as the files specify."""
    path = os.path.join(programs_dir, name + ".json")
    ins = _inputs_of(path, programs_dir)
    want = npo.run_reference(path, inputs=ins)
    got, _ = _run_gpu(path, ins)
    for k in want:
        assert got[k].dtype == want[k].dtype
        assert np.array_equal(got[k], want[k]), (name, k,
                                                 npo.max_rel_err(want[k], got[k]))


@pytest.mark.parametrize("name", [
    "jacobi2d_128x128", "jacobi3d_32x32x32_8itr_8vec", "simulator7",
    "simulator10", "varying_dimensionality"
])
def test_reference_test_programs_random_inputs(programs_dir, name):
    path = os.path.join(programs_dir, name + ".json")
    ins = _inputs_of(path, programs_dir, np.random.default_rng(SEED))
    want = npo.run_reference(path, inputs=ins)
    got, _ = _run_gpu(path, ins)
    for k in want:
        assert np.array_equal(got[k], want[k]), (name, k)


def test_run_program_driver_compare_to_reference(programs_dir, tmp_path,
                                                 monkeypatch):
    """The README call: run_program(<json>, mode, compare_to_reference=True)
    returns 0 and writes results/<name>/[reference/]<out>.dat."""
    from stencilflow_amd.run_program import run_program
    monkeypatch.chdir(tmp_path)
    path = os.path.join(programs_dir, "jacobi3d_32x32x32_8itr_8vec.json")
    ret = run_program(path, "emulation", compare_to_reference=True,
                      input_directory=programs_dir,
                      log_level=sf.LogLevel.NO_LOG)
    assert ret == 0
    out = np.fromfile(tmp_path / "results" / "jacobi3d_32x32x32_8itr_8vec" /
                      "b7.dat", np.float32)
    ref = np.fromfile(tmp_path / "results" / "jacobi3d_32x32x32_8itr_8vec" /
                      "reference" / "b7.dat", np.float32)
    assert out.size == 32**3 and np.array_equal(out, ref)


SHAPES_3D = [(32, 32, 32), (20, 44, 64), (17, 9, 12), (40, 70, 260),
             (9, 30, 512), (6, 5, 520), (8, 6, 10), (3, 3, 4), (70, 3, 8), (12, 20, 518), (10, 8, 7),
             (9, 13, 267), (5, 40, 1)]


@pytest.mark.parametrize("shape", SHAPES_3D)
@pytest.mark.parametrize("fuse", [1, 2, 3])
def test_jacobi3d_chain_random(tmp_path, shape, fuse):
    """Fused star kernel vs oracle: ragged sizes (tiles wider than the domain,
    k-tiled rows, partial j tiles, chunked i) and every fusion depth."""
    stages = 5
    rng = np.random.default_rng(SEED)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    got, desc = _run_gpu(path, {"a": x}, options={"fuse": fuse})
    # rows of 4m+2 floats run with 8-byte vectors, odd rows with single elements
    assert "star" in desc
    assert np.array_equal(got["b%d" % (stages - 1)], want), npo.max_rel_err(
        want, got["b%d" % (stages - 1)])


@pytest.mark.parametrize("options", [
    {"fuse": 2, "k1.rj": 2, "k1.by": 4},
    {"fuse": 2, "k1.rj": 3, "k1.by": 8},
    {"fuse": 2, "k1.bx": 64, "k1.by": 4, "k1.rj": 4},
    {"fuse": 2, "k1.li": 7},
    {"fuse": 4, "k1.rj": 2, "k1.by": 8},
    {"fuse": 1, "k1.li": 5},
    {"fuse": 3},
    {"fuse": 2, "k1.nt": 5},
    {"generic_only": 1},
])  # (round 5: the switches between input schemes, step orders and memory instructions are gone with their variants)
def test_jacobi3d_tile_shapes(tmp_path, options):
    shape, stages = (37, 50, 136), 4
    rng = np.random.default_rng(SEED + 1)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b3"]
    got, _ = _run_gpu(path, {"a": x}, options=options)
    assert np.array_equal(got["b3"], want)


@pytest.mark.parametrize("shape", [(20, 33, 64), (6, 5, 520), (70, 3, 8), (9, 30, 512)])
@pytest.mark.parametrize("options", [{"fuse": 1}, {"fuse": 2}, {"fuse": 3}])
def test_buffer_io_nonzero_boundary(tmp_path, shape, options):
    """Branch-free buffer loads return 0 outside a plane; a boundary constant
    other than +0 must still come out of the padding select (partial tiles,
    k-tiled rows, planes outside the domain, surplus steps of the unrolled loop)."""
    stages = 5
    rng = np.random.default_rng(SEED + 11)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages, bc_value=-0.75)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    got, desc = _run_gpu(path, {"a": x}, options=options)
    assert "star" in desc
    assert np.array_equal(got["b%d" % (stages - 1)], want)


def _lower_dim_aux_program(shape, dtype="float32"):
    """A chain whose operators also read fields lacking dimensions at the point
    itself: a [j,k] map, an [i] profile, a [k] row, a [j] column, an [i,k] sheet."""
    bc = {"type": "constant", "value": 0.5}
    return {
        "inputs": {
            "a": {"data": "constant:1.0", "data_type": dtype},
            "c2": {"data": "constant:1.0", "data_type": dtype, "input_dims": ["j", "k"]},
            "p1": {"data": "constant:1.0", "data_type": dtype, "input_dims": ["i"]},
            "r1": {"data": "constant:1.0", "data_type": dtype, "input_dims": ["k"]},
            "q1": {"data": "constant:1.0", "data_type": "float64", "input_dims": ["j"]},
            "ik": {"data": "constant:1.0", "data_type": dtype, "input_dims": ["i", "k"]},
        },
        "outputs": ["b2"],
        "dimensions": list(shape),
        "program": {
            "b0": {"computation_string": "b0 = 0.25 * (a[i-1,j,k] + a[i+1,j,k]) + c2[j,k] * a[i,j,k] + p1[i]",
                   "boundary_conditions": {"a": bc, "c2": bc, "p1": bc}, "data_type": dtype},
            "b1": {"computation_string": "b1 = b0[i,j-1,k] + b0[i,j+1,k] - r1[k] * b0[i,j,k] + q1[j]",
                   "boundary_conditions": {"b0": bc, "r1": bc, "q1": bc}, "data_type": dtype},
            "b2": {"computation_string": "b2 = (b1[i,j,k-1] + b1[i,j,k+1]) * ik[i,k]",
                   "boundary_conditions": {"b1": bc, "ik": bc}, "data_type": dtype},
        },
    }


@pytest.mark.parametrize("shape", [(12, 20, 64), (9, 7, 130), (6, 33, 37)])
@pytest.mark.parametrize("options", [{"fuse": 3}, {"fuse": 2}, {"fuse": 1}])
def test_star_chain_with_lower_dimensional_auxiliary_fields(tmp_path, shape, options):
    """Fields lacking dimensions ride along as auxiliary fields of the fused star
    kernel (indexed by the dimensions they have; one value per vector where the
    contiguous dimension is missing)."""
    prog = _lower_dim_aux_program(shape)
    rng = np.random.default_rng(SEED + 21)
    ins = {"a": rng.uniform(-1, 1, shape).astype(np.float32),
           "c2": rng.uniform(-1, 1, shape[1:]).astype(np.float32),
           "p1": rng.uniform(-1, 1, shape[:1]).astype(np.float32),
           "r1": rng.uniform(-1, 1, shape[2:]).astype(np.float32),
           "q1": rng.uniform(-1, 1, shape[1:2]),
           "ik": rng.uniform(-1, 1, (shape[0], shape[2])).astype(np.float32)}
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, inputs=ins)["b2"]
    got, desc = _run_gpu(path, ins, options=options)
    assert "[star" in desc and "[point]" not in desc, desc
    assert np.array_equal(got["b2"], want)


@pytest.mark.parametrize("shape,stages", [((40, 70, 260), 5), ((333, 264), 8)])
def test_autotuned_tile_shapes(tmp_path, shape, stages):
    """autotune=3 times three tile shapes per fused group before the first launch
    and keeps the fastest; the result does not depend on which one wins."""
    rng = np.random.default_rng(SEED + 31)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = (programs.jacobi3d if len(shape) == 3 else programs.jacobi2d)(shape, stages, bc_value=0.25)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    got, desc = _run_gpu(path, {"a": x}, options={"autotune": 3})
    assert "autotune" in desc and " -> sf_star" in desc, desc
    assert np.array_equal(got["b%d" % (stages - 1)], want)


def test_integer_bc_literal_f32_accumulation(tmp_path):
    shape = (16, 24, 64)
    rng = np.random.default_rng(SEED + 2)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, 3, bc_value=0,
                             coefficient="0.16666666666666666")
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b2"]
    got, _ = _run_gpu(path, {"a": x})
    assert np.array_equal(got["b2"], want)


@pytest.mark.parametrize("shape", [(128, 128), (40, 264), (333, 36), (64, 1024)])
@pytest.mark.parametrize("fuse", [1, 2, 4])
def test_jacobi2d_chain_random(tmp_path, shape, fuse):
    stages = 6
    rng = np.random.default_rng(SEED + 3)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi2d(shape, stages, bc_value=1.0)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b5"]
    got, desc = _run_gpu(path, {"a": x}, options={"fuse": fuse})
    assert "star" in desc
    assert np.array_equal(got["b5"], want)


@pytest.mark.parametrize("fuse", [1, 3])
def test_three_operator_chain_f64(tmp_path, fuse):
    """C5 at a size the oracle finishes quickly: diffusion -> advection ->
    laplacian, float64, scalar coefficient inputs."""
    shape = (24, 40, 136)
    rng = np.random.default_rng(SEED + 4)
    prog = programs.diffusion_advection_laplacian(shape)
    x = rng.uniform(-1, 1, shape)
    ins = _inputs_of(prog)
    ins["a"] = x
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, inputs=ins)["lap"]
    got, desc = _run_gpu(path, ins, options={"fuse": fuse})
    assert np.array_equal(got["lap"], want)
    if fuse == 3:
        assert "1 launches" in desc and "T=3" in desc


def test_shrink_boundary(tmp_path):
    shape = (12, 16, 32)
    rng = np.random.default_rng(SEED + 5)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, 2)
    for k in prog["program"].values():
        for bc in k["boundary_conditions"].values():
            bc["type"] = "shrink"
            del bc["value"]
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b1"]
    got, _ = _run_gpu(path, {"a": x})
    assert np.array_equal(got["b1"], want)
    h = 2  # outside the shrink halo the junk never arrives
    inner = tuple(slice(h, -h) for _ in shape)
    assert np.abs(got["b1"][inner]).max() <= 1.0


def test_math_calls_within_tolerance(tmp_path):
    prog = {
        "inputs": {"a": {"data": "constant:1.0", "data_type": "float32"}},
        "outputs": ["c"],
        "dimensions": [8, 16, 32],
        "program": {
            "b": {
                "computation_string":
                "b = sin(a[i,j,k]) * cos(a[i,j,k+1]) + sqrt(fabs(a[i-1,j,k]))",
                "boundary_conditions": {"a": {"type": "constant", "value": 0.5}},
                "data_type": "float32"
            },
            "c": {
                "computation_string":
                "t = max(b[i,j,k], b[i,j-1,k]); c = t if t > 0.3 else min(t, 0.1) - exp(b[i,j,k])",
                "boundary_conditions": {"b": {"type": "constant", "value": 0.0}},
                "data_type": "float32"
            }
        }
    }
    rng = np.random.default_rng(SEED + 6)
    x = rng.uniform(-1, 1, (8, 16, 32)).astype(np.float32)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["c"]
    got, _ = _run_gpu(path, {"a": x})
    assert npo.arrays_match(want, got["c"], 1e-6)  # tolerance: BASELINE.json


def test_full_size_properties():
    """BASELINE size (512^3, f32): properties that do not need the oracle to run
    the whole chain -- fused == unfused bit for bit, 8-fold symmetry, range --
    plus a direct oracle check of the first two stages."""
    n, stages = 512, 6
    prog = programs.jacobi3d((n, n, n), stages)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "big.json"))
        x = np.ones((n, n, n), np.float32)
        fused, _ = _run_gpu(path, {"a": x}, options={"fuse": 2})
        unfused, _ = _run_gpu(path, {"a": x}, options={"fuse": 1})
    out = fused["b5"]
    assert np.array_equal(out, unfused["b5"])
    assert np.array_equal(out, out[::-1]) and np.array_equal(out, out[:, ::-1])
    assert np.array_equal(out, out[:, :, ::-1])
    assert np.array_equal(out, out.transpose(1, 0, 2))
    assert 0.0 <= out.min() and out.max() <= 1.0
    # corner block against the oracle on a sub-domain whose far faces are
    # > `stages` cells away from the compared region
    m = 40
    small = programs.jacobi3d((m, m, m), stages)
    ref = npo.run_reference(small, {"a": np.ones((m, m, m), np.float32)})["b5"]
    c = m - stages - 1
    assert np.array_equal(out[:c, :c, :c], ref[:c, :c, :c])


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("groups", [1, 2, 4])
@pytest.mark.parametrize("early", [False, True])
def test_slab_decomposition_in_process(tmp_path, world, overlap, groups, early):
    """Slab-decomposed execution (one plan per rank, halos copied between the
    ranks' device buffers by the driver) equals the undivided run bit for bit.
    All ranks share this GPU; the transport is the only part not covered."""
    from stencilflow_amd.distributed import (LocalExchanger, SlabRunner,
                                             run_lockstep)
    from stencilflow_amd.lowering import lower
    shape, stages = (52, 24, 64), 11
    rng = np.random.default_rng(SEED + 7)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages, bc_value=0.5)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b10"]
    sfir = lower(sf.KernelChainGraph(path))
    exch = LocalExchanger(world)
    views = [exch.for_rank(r) for r in range(world)]
    if groups == 2:
        # as with a device-side (RCCL) exchange: the overlapped interior launch
        # leaves compute units free, i.e. is chunked differently -- same results
        for v in views:
            v.reserved_cus = 32
    runners = [SlabRunner(sfir, shape, r, world, options={"fuse": 2},
                          exchanger=views[r], overlap=overlap,
                          groups_per_exchange=groups, early_exchange=early)
               for r in range(world)]
    assert runners[0].is_chain and runners[0].halo == 2 * groups
    for r in runners:
        r.upload([x[r.lo:r.hi]])
    run_lockstep(runners)
    got = np.zeros(shape, np.float32)
    for r in runners:
        part = np.zeros(r.local_shape, np.float32)
        r.download([part])
        got[r.lo:r.hi] = part
        r.close()
    assert np.array_equal(got, want)


def test_long_chain(tmp_path):
    """300 operators -> 150 launches of one compiled kernel, ping-pong buffers
    (regression: every launch of a long schedule must be planned)."""
    shape, stages = (24, 20, 32), 300
    rng = np.random.default_rng(SEED + 8)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    prog = programs.jacobi3d(shape, stages, bc_value=1.0)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": x})["b299"]
    got, desc = _run_gpu(path, {"a": x})
    assert "150 launches" in desc and "4 device buffers" in desc
    assert np.array_equal(got["b299"], want)


def test_slab_dag_program_in_process(tmp_path):
    """A DAG (two inputs, fan-out, generic kernels) under slab decomposition:
    every slab-split field a launch reads across planes is exchanged."""
    from stencilflow_amd.distributed import (LocalExchanger, SlabRunner,
                                             run_lockstep)
    from stencilflow_amd.lowering import lower
    shape = (20, 12, 16)
    prog = {
        "inputs": {"a": {"data": "constant:1.0", "data_type": "float32"},
                   "w": {"data": "constant:1.0", "data_type": "float64"}},
        "outputs": ["e", "d"],
        "dimensions": list(shape),
        "program": {
            "b": {"computation_string": "b = a[i-1,j,k] + w[i+2,j,k-1] * 0.5",
                  "boundary_conditions": {"a": {"type": "constant", "value": 1.0},
                                          "w": {"type": "constant", "value": -1.0}},
                  "data_type": "float64"},
            "d": {"computation_string": "d = b[i,j+1,k+1] * a[i,j,k] - b[i+1,j,k]",
                  "boundary_conditions": {"b": {"type": "constant", "value": 0.25}},
                  "data_type": "float32"},
            "e": {"computation_string": "e = d[i-1,j,k] + b[i,j,k] + w[i,j,k]",
                  "boundary_conditions": {"d": {"type": "constant", "value": 2.0}},
                  "data_type": "float64"}}}
    rng = np.random.default_rng(SEED + 9)
    a = rng.uniform(-1, 1, shape).astype(np.float32)
    w = rng.uniform(-1, 1, shape)
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, {"a": a, "w": w})
    sfir = lower(sf.KernelChainGraph(path))
    world = 2
    exch = LocalExchanger(world)
    runners = [SlabRunner(sfir, shape, r, world, exchanger=exch.for_rank(r))
               for r in range(world)]
    assert not runners[0].is_chain
    for r in runners:
        r.upload([a[r.lo:r.hi], w[r.lo:r.hi]])
    run_lockstep(runners)
    got_e = np.zeros(shape, np.float64)
    got_d = np.zeros(shape, np.float32)
    for r in runners:
        pe = np.zeros(r.local_shape, np.float64)
        pd = np.zeros(r.local_shape, np.float32)
        outs = {"e": pe, "d": pd}
        r.download([outs[n] for n in r.plan.output_names])
        got_e[r.lo:r.hi], got_d[r.lo:r.hi] = pe, pd
        r.close()
    assert np.array_equal(got_d, want["d"]) and np.array_equal(got_e, want["e"])


@pytest.mark.parametrize("world", [2, 3])
def test_slab_star_chain_with_lower_dimensional_auxiliary_fields(tmp_path, world):
    """The same under slab decomposition: [i] and [i,k] fields are split with the
    slabs, [j,k], [k] and [j] fields are whole on every rank."""
    from stencilflow_amd.distributed import (LocalExchanger, SlabRunner, run_lockstep)
    from stencilflow_amd.lowering import lower
    shape = (30, 14, 66)
    prog = _lower_dim_aux_program(shape)
    rng = np.random.default_rng(SEED + 22)
    ins = {"a": rng.uniform(-1, 1, shape).astype(np.float32),
           "c2": rng.uniform(-1, 1, shape[1:]).astype(np.float32),
           "p1": rng.uniform(-1, 1, shape[:1]).astype(np.float32),
           "r1": rng.uniform(-1, 1, shape[2:]).astype(np.float32),
           "q1": rng.uniform(-1, 1, shape[1:2]),
           "ik": rng.uniform(-1, 1, (shape[0], shape[2])).astype(np.float32)}
    path = _write(tmp_path, prog)
    want = npo.run_reference(prog, inputs=ins)["b2"]
    sfir = lower(sf.KernelChainGraph(path))
    exch = LocalExchanger(world)
    runners = [SlabRunner(sfir, shape, r, world, exchanger=exch.for_rank(r)) for r in range(world)]
    assert "[star" in runners[0].plan.describe()
    split = {"a", "p1", "ik"}  # fields with the outermost dimension
    for r in runners:
        r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi]) if n in split else ins[n]
                  for n in r.plan.input_names])
    run_lockstep(runners)
    got = np.zeros(shape, np.float32)
    for r in runners:
        part = np.zeros(r.local_shape, np.float32)
        r.download([part])
        got[r.lo:r.hi] = part
        r.close()
    assert np.array_equal(got, want)


def test_full_size_jacobi2d_properties():
    """BASELINE configs[1] size (4096 x 4096 f32): fused == unfused bit for bit,
    symmetry of the constant-input solution, corner block against the oracle."""
    import tempfile
    n, stages = 4096, 12
    prog = programs.jacobi2d((n, n), stages)
    x = np.ones((n, n), np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "c2.json"))
        fused, desc = _run_gpu(path, {"a": x}, options={"fuse": 4})
        unfused, _ = _run_gpu(path, {"a": x}, options={"fuse": 1})
    assert "star2d" in desc and "T=4" in desc
    out = fused["b11"]
    assert np.array_equal(out, unfused["b11"])
    assert np.array_equal(out, out.T) and np.array_equal(out, out[::-1])
    assert np.array_equal(out, out[:, ::-1])
    assert 0.0 <= out.min() and out.max() <= 1.0
    m = 64
    ref = npo.run_reference(programs.jacobi2d((m, m), stages),
                            {"a": np.ones((m, m), np.float32)})["b11"]
    c = m - stages - 1
    assert np.array_equal(out[:c, :c], ref[:c, :c])


def test_full_size_three_operator_chain_properties():
    """BASELINE configs[4] size (512^3 f64): the chain fused into ONE launch
    equals the three separate launches bit for bit; a corner block with random
    data equals the oracle."""
    import tempfile
    n = 512
    prog = programs.diffusion_advection_laplacian((n, n, n))
    rng = np.random.default_rng(SEED + 10)
    x = rng.random((n, n, n))
    ins = _inputs_of(prog)
    ins["a"] = x
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "c5.json"))
        fused, desc = _run_gpu(path, ins)
        assert "1 launches" in desc and "T=3" in desc and "f64" in desc
        split, desc1 = _run_gpu(path, ins, options={"fuse": 1})
        assert "3 launches" in desc1
    assert np.array_equal(fused["lap"], split["lap"])
    del split
    m = 24
    small = programs.diffusion_advection_laplacian((m, m, m))
    sins = _inputs_of(small)
    sins["a"] = np.ascontiguousarray(x[:m, :m, :m])
    ref = npo.run_reference(small, inputs=sins)["lap"]
    c = m - 4
    assert np.array_equal(fused["lap"][:c, :c, :c], ref[:c, :c, :c])


@pytest.mark.parametrize("shape,stages,options,launch", [
    ((6, 10, 16), 1, None, "[star"),
    ((6, 10, 16), 1, {"generic_only": 1}, "[point"),
    ((21, 24, 64), 5, {"fuse": 2}, "[star T=2"),
    ((9, 12, 520), 4, {"fuse": 3}, "[star T=3"),
    ((70, 136), 6, None, "[star"),
    ((70, 136), 3, {"generic_only": 1}, "[point"),
])
def test_copy_boundary_condition(tmp_path, shape, stages, options, launch):
    """`copy`: out-of-domain reads take the centre value (the FPGA expansions'
    semantics, reference stencil/intel_fpga.py:225-227; the reference's CPU
    path raises NameError for it, stencil/cpu.py:87 -- so this is checked
    against a direct NumPy statement of that rule, not against the oracle):
    for a star of radius 1 that is the field padded with its own edge values.
    Fused star kernels (any depth, 3-D and 2-D) and the generic kernel."""
    if len(shape) == 3:
        prog = programs.jacobi3d(shape, stages)
    else:
        prog = programs.jacobi2d(shape, stages)
    for k in prog["program"].values():
        for f in k["boundary_conditions"]:
            k["boundary_conditions"][f] = {"type": "copy"}
    rng = np.random.default_rng(SEED + 11)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    path = _write(tmp_path, prog)
    got, desc = _run_gpu(path, {"a": x}, options=options)
    assert launch in desc, desc
    if launch != "[point":
        assert "[point" not in desc
    # no access of these operators carries a literal (copy keeps the field type): the sum is
    # accumulated in float32, in the order of the program text; the coefficient is a double
    want = x
    for _ in range(stages):
        p = np.pad(want, 1, mode="edge")
        c = tuple(slice(1, -1) for _ in shape)

        def nb(axis, off):
            idx = list(c)
            idx[axis] = slice(1 + off, p.shape[axis] - 1 + off)
            return p[tuple(idx)]

        if len(shape) == 3:
            s = ((((nb(0, -1) + nb(0, 1)) + nb(1, -1)) + nb(1, 1)) + nb(2, -1)) + nb(2, 1)
            want = (0.16666666 * s.astype(np.float64)).astype(np.float32)
        else:
            s = ((nb(0, -1) + nb(0, 1)) + nb(1, -1)) + nb(1, 1)
            want = (0.25 * s.astype(np.float64)).astype(np.float32)
    name = "b{}".format(stages - 1)
    assert np.array_equal(got[name], want), npo.max_rel_err(want, got[name])


def _copy_read(x, off):
    """x read at p + off, points whose read leaves the domain taking x[p] (the `copy` rule)."""
    valid = np.ones(x.shape, bool)
    src = x
    for axis, o in enumerate(off):
        if o == 0:
            continue
        src = np.roll(src, -o, axis=axis)
        idx = [slice(None)] * x.ndim
        idx[axis] = slice(x.shape[axis] - o, None) if o > 0 else slice(0, -o)
        valid[tuple(idx)] = False
    return np.where(valid, src, x)


_CROSS2 = [(0, 0, 0)] + [tuple(s * d if a == ax else 0 for a in range(3)) for ax in range(3) for d in (1, 2) for s in (-1, 1)]
_BOX1 = [(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)]
_SCATTER2 = [(0, 0, 0), (-2, 1, 0), (2, -2, 1), (1, 1, -2), (0, 2, 2), (-1, -2, -1), (2, 0, 0), (0, -1, 2), (-2, -2, -2), (1, 0, 1)]
_BOX2_2D = [(j, k) for j in (-2, -1, 0, 1, 2) for k in (-2, -1, 0, 1, 2)]


@pytest.mark.parametrize("offsets,shape,dtype,stages,options,launch", [
    (_CROSS2, (11, 14, 72), "float32", 4, {"fuse": 2}, "[wide star T=2"),
    (_CROSS2, (9, 10, 40), "float64", 2, None, "[wide star"),
    (_BOX1, (10, 13, 68), "float32", 4, {"fuse": 2}, "[compact"),
    (_SCATTER2, (12, 21, 72), "float32", 2, None, "[dense"),
    (_BOX2_2D, (45, 136), "float32", 3, None, "[dense"),
    (_SCATTER2, (6, 8, 16), "float64", 2, {"generic_only": 1}, "[point"),
])
def test_copy_boundary_in_the_other_fused_kernels(tmp_path, offsets, shape, dtype, stages, options, launch):
    """`copy` in the wide-star, compact and dense kernels (and, last row, the generic kernel on the same
    kind of operator): every read that leaves the domain -- along any of its offsets -- takes the value at
    the point itself; NumPy statement of the rule, sum in the order of the text, in the field's type."""
    its = ["i", "j", "k"][3 - len(shape):]
    np_t = np.float32 if dtype == "float32" else np.float64

    def expr(f):
        terms = ["{}[{}]".format(f, ",".join(it if o == 0 else "{}{:+d}".format(it, o) for it, o in zip(its, off)))
                 for off in offsets]
        return "0.0625 * (" + " + ".join(terms) + ")"

    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": ["b{}".format(stages - 1)],
            "dimensions": list(shape), "program": {}}
    prev = "a"
    for s in range(stages):
        name = "b{}".format(s)
        prog["program"][name] = {"computation_string": "{} = {}".format(name, expr(prev)),
                                 "boundary_conditions": {prev: {"type": "copy"}}, "data_type": dtype}
        prev = name
    x = np.random.default_rng(SEED + 12).uniform(-1, 1, shape).astype(np_t)
    got, desc = _run_gpu(_write(tmp_path, prog), {"a": x}, options=options)
    assert launch in desc, desc
    if launch != "[point":
        assert "[point" not in desc
    want = x
    for _ in range(stages):
        s = _copy_read(want, offsets[0])
        for off in offsets[1:]:
            s = s + _copy_read(want, off)
        want = (0.0625 * s.astype(np.float64)).astype(np_t)
    assert np.array_equal(got[prev], want), npo.max_rel_err(want, got[prev])


@pytest.mark.parametrize("args,kwargs", [
    (("float32", 4, 0, 12, 20, 32, 1, 1, 1), {}),
    (("float32", 3, 0, 10, 12, 16, 1, 1, 1), {"stencil_shape": "diffusion"}),
    (("float64", 3, 1, 40, 36, 0, 2, 1, 0), {"stencil_shape": "box"}),
    (("float32", 5, 0.5, 24, 32, 0, 1, 2, 0), {"fork_frequency": 0.5,
                                              "fork_length_left": 1,
                                              "fork_length_right": 2}),
    (("float64", 3, 0, 8, 12, 16, 1, 1, 1), {"stencil_shape": "hotspot"}),
    (("float32", 3, 0.4, 28, 40, 0, 1, 1, 0), {"stencil_shape": "hotspot"}),
    (("float32", 6, 0, 200, 0, 0, 3, 0, 0), {}),
])
def test_synthesized_programs(tmp_path, args, kwargs):
    """Programs from the workload generator (the reference's bin/synthesize.py
    conventions): cross / box / diffusion / hotspot shapes, extra input fields,
    forks and joins, 1-D to 3-D -- random array inputs, oracle comparison."""
    prog, name = programs.synthesize(*args, **kwargs)
    path = _write(tmp_path, prog, name[:-5])
    ins = _inputs_of(path, None, np.random.default_rng(SEED + 12))
    want = npo.run_reference(path, inputs=ins)
    got, _ = _run_gpu(path, ins)
    for k in want:
        assert np.array_equal(got[k], want[k]), (name, k, npo.max_rel_err(want[k], got[k]))


def test_run_program_flags(programs_dir, tmp_path, monkeypatch):
    """Driver flags of the reference (bin/run_program.py:12-37): -generate-input
    (all inputs constant 0.5), -repetitions, -halo pruning of the saved files,
    skip-execution returning None, unknown mode raising ValueError."""
    from stencilflow_amd.run_program import run_program
    monkeypatch.chdir(tmp_path)
    prog = programs.jacobi3d((12, 16, 32), 3)
    for k in prog["program"].values():
        for bc in k["boundary_conditions"].values():
            bc["type"] = "shrink"
            del bc["value"]
    path = _write(tmp_path, prog, "shrunk")
    assert run_program(path, "hardware", compare_to_reference=True, generate_input=True,
                       halo=3, repetitions=2, log_level=sf.LogLevel.NO_LOG) == 0
    out = np.fromfile(tmp_path / "results" / "shrunk" / "b2.dat", np.float32)
    assert out.size == 6 * 10 * 26          # pruned by `halo` on every side
    ref = npo.run_reference(prog, generate_input=True)["b2"][3:-3, 3:-3, 3:-3]
    assert np.array_equal(out, ref.ravel()) and len(np.unique(out)) == 1
    assert run_program(path, "emulation", skip_execution=True,
                       log_level=sf.LogLevel.NO_LOG) is None
    assert run_program(path, "hip", log_level=sf.LogLevel.NO_LOG) is None
    with pytest.raises(ValueError, match="Unrecognized execution mode"):
        run_program(path, "simulation", log_level=sf.LogLevel.NO_LOG)


@pytest.mark.parametrize("graph", [0, 1])
def test_graph_replay_follows_scalar_changes(tmp_path, graph):
    """Launch-bound chains are replayed as one hipGraph (csrc/exec.cpp: execute);
    the captured launches carry the scalar arguments by value, so changing a
    scalar input must rebuild the graph.  Same plan, three runs, two scalar
    sets; graph=0 is the plain stream path."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    shape = (12, 20, 32)
    rng = np.random.default_rng(SEED + 21)
    prog = programs.diffusion_advection_laplacian(shape, repeats=3)  # 9 operators
    x = rng.uniform(-1, 1, shape)
    path = _write(tmp_path, prog)
    chain = sf.KernelChainGraph(path)
    out_name = prog["outputs"][0]
    with Plan(lower(chain), options={"graph": graph, "fuse": 2}) as plan:
        assert plan.num_launches >= 4
        for scale in (1.0, 0.5, 1.0):
            ins = _inputs_of(prog)
            for k in ins:
                ins[k] = ins[k] * scale
            ins["a"] = x
            want = npo.run_reference(prog, inputs=ins)[out_name]
            plan.set_scalars([ins[n] for n in plan.scalar_names])
            got = np.zeros(shape)
            plan.run([x], [got], 2)  # two repetitions on the same buffers
            assert np.array_equal(got, want), (graph, scale)


@pytest.mark.parametrize("seed", range(0, 16))
def test_random_programs_under_slab_decomposition(tmp_path, seed):
    """tools/slab_fuzz.py: a random star chain or DAG, 2-4 ranks with unequal
    slabs, random halo depth and overlap switch, all ranks in this process."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "slab_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                  "tools", "slab_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    status, detail = mod.run_seed(seed, str(tmp_path))
    _SLAB_FUZZ_STATUS[seed] = status
    if status == "skip":
        # not a pass: the seed drew a program that cannot be decomposed (1-D, or slabs thinner than the halo);
        # reported by name, and bounded by the test that follows
        pytest.skip("slab_fuzz seed {}: {}".format(seed, detail or "a 1-D program has no slab axis to split"))
    assert status == "ok", detail


_SLAB_FUZZ_STATUS = {}


def test_slab_fuzz_skips_are_counted_and_bounded():
    """VERDICT r03 (weak 3): a skipped seed used to pass silently.  At most 2 of the 16 seeds above may skip,
    and the ones that do are printed."""
    if len(_SLAB_FUZZ_STATUS) < 16:
        pytest.skip("runs after the 16 seeds of test_random_programs_under_slab_decomposition (a -k selection left some out)")
    skipped = sorted(s for s, st in _SLAB_FUZZ_STATUS.items() if st == "skip")
    print("slab_fuzz: {} of 16 seeds skipped: {}".format(len(skipped), skipped))
    assert len(skipped) <= 2, skipped
    assert sum(1 for st in _SLAB_FUZZ_STATUS.values() if st == "ok") >= 14


def test_full_benchmark_configuration_bit_exact():
    """C3 itself -- jacobi3d 512^3 float32, the 1000-operator chain on random
    data -- against the C oracle, every one of the 134 million results compared
    bit for bit.  The oracle applies an 8-operator program 125 times, feeding
    each result back (all operators of the chain are the same operator), which
    is the same arithmetic as the 1000-operator program."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    n, stages, block = 512, 1000, 8
    rng = np.random.default_rng(SEED + 31)
    x = rng.uniform(0, 1, (n, n, n)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d((n, n, n), stages), os.path.join(tmp, "c3.json"))
        chain = sf.KernelChainGraph(path)
    got = np.zeros((n, n, n), np.float32)
    with Plan(lower(chain)) as plan:
        # (round 5: 332 launches of three operators in the dense kernel's fused streaming form, the last four
        #  operators two by two on the star kernel -- the plan `python bench.py` times)
        assert "[dense T=3 block 34x30 rows/thread 1" in plan.describe() and plan.num_launches == 334, plan.describe()[:400]
        plan.run([x], [got], 1)
    # ... and the star kernel's plan of rounds 1-4, two operators per launch, which dense.t2=0 still selects
    got2 = np.zeros((n, n, n), np.float32)
    with Plan(lower(chain), options={"dense.t2": 0}) as plan:
        assert "star T=2" in plan.describe() and plan.num_launches == stages // 2
        plan.run([x], [got2], 1)
    assert np.array_equal(got, got2)
    ref = c_oracle.CompiledReference(programs.jacobi3d((n, n, n), block))
    ref.threads = _oracle_threads()
    want = x
    for _ in range(stages // block):
        want = ref.run({"a": want})["b%d" % (block - 1)]
    assert want.dtype == got.dtype
    assert np.array_equal(got, want), "max |diff| = %g" % float(np.max(np.abs(got - want)))
    assert float(got.max()) > 0.0  # not a field of zeros


def _oracle_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, 2 * (-(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def test_full_c2_configuration_bit_exact():
    """C2 itself: jacobi2d 4096^2 float32, 1000 operators, random data, all
    16.8 million results against the C oracle (8 operators x 125, fed back)."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    n, stages, block = 4096, 1000, 8
    x = np.random.default_rng(SEED + 32).uniform(0, 1, (n, n)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi2d((n, n), stages), os.path.join(tmp, "c2.json"))
        chain = sf.KernelChainGraph(path)
    got = np.zeros((n, n), np.float32)
    with Plan(lower(chain)) as plan:
        assert "star T=4" in plan.describe()
        plan.run([x], [got], 1)
    ref = c_oracle.CompiledReference(programs.jacobi2d((n, n), block))
    ref.threads = _oracle_threads()
    want = x
    for _ in range(stages // block):
        want = ref.run({"a": want})["b%d" % (block - 1)]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dims,dtype,bc,sum_form", [
    ((20, 24, 72), "float32", {"type": "constant", "value": 0}, True),
    ((11, 37, 40), "float32", {"type": "constant", "value": 0.5}, True),    # float literal: the sum runs in double
    ((9, 14, 24), "float64", {"type": "constant", "value": 0.25}, True),
    ((12, 18, 40), "float32", {"type": "shrink"}, True),
    ((70, 136), "float32", {"type": "constant", "value": 0.0}, True),
])
def test_generator_box_of_extent_two_in_the_plain_sum_form(tmp_path, dims, dtype, bc, sum_form):
    """The generator's box of extent 2 (125 points, 25 in 2-D) is ONE left-associated sum whose terms come plane by
    plane: the dense kernel reads every plane from LDS once -- where it arrived by LDS-DMA, round 5 -- and adds it to
    the five output planes that are open (SF_DENSE_STREAM; codegen.hpp: dense_sum_form, stream_schedule).  Same
    results, bit for bit, as the oracle -- whatever type the boundary literal gives the sum."""
    full = list(dims) + [0] * (3 - len(dims))
    ext = [2 if d else 0 for d in full]
    prog, _ = programs.synthesize(dtype, 2, 0.0, *full, *ext, stencil_shape="box")
    for k in prog["program"].values():
        for f in k["boundary_conditions"]:
            k["boundary_conditions"][f] = dict(bc)
    x = np.random.default_rng(SEED + 35).uniform(-1, 1, dims).astype(dtype)
    path = _write(tmp_path, prog)
    chain = sf.KernelChainGraph(path)
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    got = np.zeros(dims, dtype)
    with Plan(lower(chain)) as plan:
        assert "[dense" in plan.describe(), plan.describe()
        assert ("#define SF_DENSE_STREAM 1" in plan.kernel_source(0)) == sum_form
        plan.run([x], [got], 1)
    if bc["type"] == "copy":
        with Plan(lower(chain), options={"generic_only": 1}) as ref:
            want = np.zeros(dims, dtype)
            ref.run([x], [want], 1)
    else:
        want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    if bc["type"] == "shrink":
        inner = tuple(slice(4, -4) for _ in dims)
        assert np.array_equal(got[inner], want[inner])
    else:
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("dims,dtype,bc", [
    ((22, 30, 72), "float32", {"type": "constant", "value": 0}),
    ((11, 17, 24), "float64", {"type": "constant", "value": 0.25}),
    ((15, 19, 40), "float32", {"type": "shrink"}),
    ((90, 136), "float32", {"type": "constant", "value": -1}),
    ((60, 72), "float32", {"type": "constant", "value": 0.5}),       # float literal: the sum runs in double (49 points: still fits)
    ((10, 13, 24), "float32", {"type": "constant", "value": 0.5}),   # ... and 343 points: one row per thread (round 5)
])  # (a 343-term operator takes ~25 s to compile per kernel form: few cases, the generic cross-check on the 2-D ones only)
def test_generator_box_of_extent_three_streams_through_the_dense_kernel(tmp_path, dims, dtype, bc):
    """The generator's box of extent 3 (343 points, 49 in 2-D; verdict r03, next 9: radius 3) is a plain sum ordered by
    plane like its smaller siblings: the dense kernel's streaming form with seven open output planes and four halo
    columns per LDS row (codegen.hpp: dense_r3_eligible).  Same results, bit for bit, as the oracle; dense=0 leaves
    it to the generic kernel, which must agree."""
    full = list(dims) + [0] * (3 - len(dims))
    ext = [3 if d else 0 for d in full]
    prog, _ = programs.synthesize(dtype, 2, 0.0, *full, *ext, stencil_shape="box")
    for k in prog["program"].values():
        for f in k["boundary_conditions"]:
            k["boundary_conditions"][f] = dict(bc)
    x = np.random.default_rng(SEED + 36).uniform(-1, 1, dims).astype(dtype)
    chain = sf.KernelChainGraph(_write(tmp_path, prog))
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    got = np.zeros(dims, dtype)
    # (a float boundary literal types the sum of a float32 operator double: seven sets of two-register accumulators.
    #  Until round 4 they spilled in every 3-D tile shape and the operator ran on the generic kernel, 46 x slower; with
    #  the planes arriving by LDS-DMA -- no staging registers -- the one-row-per-thread shape is clean)
    with Plan(lower(chain)) as plan:
        assert "[dense" in plan.describe(), plan.describe()
        src = plan.kernel_source(0)
        assert "#define SF_R 3" in src and "#define SF_RC 4" in src and "#define SF_ACCS 7" in src
        assert "#define SF_IN_SLOTS" in src  # (planes by LDS-DMA)
        plan.run([x], [got], 1)
    ref = None
    if len(dims) == 2:
        with Plan(lower(chain), options={"dense": 0}) as plan:
            assert "[dense" not in plan.describe()
            ref = np.zeros(dims, dtype)
            plan.run([x], [ref], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    if ref is None:
        ref = want
    if bc["type"] == "shrink":
        inner = tuple(slice(6, -6) for _ in dims)
        assert np.array_equal(got[inner], want[inner]) and np.array_equal(ref[inner], want[inner])
    else:
        assert np.array_equal(got, want, equal_nan=True) and np.array_equal(ref, want, equal_nan=True)


@pytest.mark.parametrize("args,kwargs,stages,kernel,options", [
    ((2, 2, 2), {}, 4, "sf_dense3d_f32_t2_", None),                               # round 5: two radius-2 crosses per streaming dense launch
    ((2, 2, 2), {}, 4, "[wide star T=2", {"dense.t2": 0}),                        # the wide-star kernel (rounds 3-4)
    ((1, 1, 1), {"stencil_shape": "box"}, 4, "sf_dense3d_f32_t2_", None),         # round 4: two boxes per streaming dense launch
    ((1, 1, 1), {"stencil_shape": "box"}, 4, "[compact", {"dense.t2": 0}),        # the compact kernel (rounds 2-3)
    ((2, 2, 2), {"stencil_shape": "box"}, 2, "[dense", None),
])
def test_full_size_generator_workloads_bit_exact(args, kwargs, stages, kernel, options):
    """The reference generator's radius-2 cross, 27-point box and 125-point box (bin/synthesize.py
    conventions; the first two are bench.py's `wide` and `box` workloads) at the benchmark's 512^3,
    random data: all 134 million results of a short chain against the C oracle -- the tile and
    chunk shapes the planner picks for the full-size grid, which small random programs never see."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    n = 512
    prog, _ = programs.synthesize("float32", stages, 0.0, n, n, n, *args, **kwargs)
    x = np.random.default_rng(SEED + 34).uniform(-1, 1, (n, n, n)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "w.json")))
    out = prog["outputs"][0]
    got = np.zeros((n, n, n), np.float32)
    with Plan(lower(chain), options=options) as plan:
        assert kernel in plan.describe(), plan.describe()
        assert list(plan.output_names) == [out] and list(plan.input_names) == ["a"]
        plan.run([x], [got], 1)
    ref = c_oracle.CompiledReference(prog)
    ref.threads = _oracle_threads()
    want = ref.run({"a": x})[out]
    assert np.array_equal(got, want), npo.max_rel_err(want, got)


@pytest.mark.parametrize("dims,options,expect", [
    ((512, 512, 512), {"dag": 0}, "14 launches"),               # depth-first order: each branch a chain of its own
    ((512, 512, 512), None, "[dag: 4 stages, 3 windows]"),       # both branches of a fork from one read
    ((4096, 4096), None, "[dag: 6 stages"),                      # 2-D: fork, branches, join and what follows in one launch
])
def test_full_size_fork_join_programs_bit_exact(dims, options, expect):
    """VERDICT r03 (next 2): the reference generator's fork / join program (`synthesize float32 16 0 ... 1 1 1
    -fork_frequency 0.25`: 28 operators -- chains, two branches of two operators every fourth stage, a two-field
    operator joining them; bin/synthesize.py:228-253) at full size on random data, all results against the C
    oracle.  3-D: the depth-first operator order (each branch fuses like a chain), and with a third register
    window the sibling groups (both branches of a fork evaluated from one read of the forked field, two fields
    materialised); 2-D: DAG groups that hold a fork, its branches, the join and the chain after it."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    size = list(dims) + [0] * (3 - len(dims))
    ext = [1] * len(dims) + [0] * (3 - len(dims))
    prog, _ = programs.synthesize("float32", 16, 0.0, size[0], size[1], size[2], ext[0], ext[1], ext[2], fork_frequency=0.25)
    assert len(prog["program"]) == 28
    x = np.random.default_rng(SEED + 41).uniform(-1, 1, dims).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "fork.json")))
    out = prog["outputs"][0]
    got = np.zeros(dims, np.float32)
    with Plan(lower(chain), options=options) as plan:
        assert expect in plan.describe(), plan.describe()
        plan.run([x], [got], 1)
    ref = c_oracle.CompiledReference(prog)
    ref.threads = _oracle_threads()
    want = ref.run({"a": x})[out]
    assert np.array_equal(got, want), npo.max_rel_err(want, got)


def test_full_c5_configuration_bit_exact():
    """C5 itself: diffusion -> advection -> laplacian on 512^3 float64 random
    data, fused into one launch, all 134 million results against the C oracle."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    n = 512
    prog = programs.diffusion_advection_laplacian((n, n, n))
    x = np.random.default_rng(SEED + 33).uniform(-1, 1, (n, n, n))
    ins = _inputs_of(prog)
    ins["a"] = x
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "c5.json"))
        chain = sf.KernelChainGraph(path)
    got = np.zeros((n, n, n), np.float64)
    with Plan(lower(chain)) as plan:
        assert "1 launches" in plan.describe() and "star T=3" in plan.describe()
        plan.set_scalars([ins[k] for k in plan.scalar_names])
        plan.run([x], [got], 1)
    ref = c_oracle.CompiledReference(prog)
    ref.threads = _oracle_threads()
    want = ref.run(ins)["lap"]
    assert np.array_equal(got, want)


C4_SHAPE, C4_STAGES = (4096, 512, 512), 120
_C4 = {}


def c4_input(lo=0, hi=C4_SHAPE[0]):
    """Planes [lo, hi) of the C4 test grid: 512-plane blocks seeded 4000 + block (every rank of
    a decomposed run can make its own slab)."""
    parts = []
    for b in range(lo // 512, -(-hi // 512)):
        block = np.random.default_rng(4000 + b).random((512, ) + C4_SHAPE[1:], dtype=np.float32)
        parts.append(block[max(lo, 512 * b) - 512 * b:min(hi, 512 * (b + 1)) - 512 * b])
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


def c4_oracle():
    """The C oracle's result for the C4 grid after C4_STAGES operators, computed once per test
    session (the in-process and the multi-process C4 tests compare with the same array)."""
    if "want" not in _C4:
        from oracle import c_oracle
        block = 8
        ref = c_oracle.CompiledReference(programs.jacobi3d(C4_SHAPE, block))
        ref.threads = _oracle_threads()
        want = c4_input()
        for _ in range(C4_STAGES // block):
            want = ref.run({"a": want})["b%d" % (block - 1)]
        _C4["want"] = want
    return _C4["want"]


def test_full_c4_grid_eight_slabs_bit_exact():
    """C4's grid -- 4096 x 512 x 512 float32 split into eight 512-plane slabs,
    deep halos, overlapped exchange with reserved compute units -- for the
    first 120 operators of the chain, all 1.07 billion results against the C
    oracle.  The eight ranks share this GPU; halos are copied by the in-process
    exchanger (tests/test_distributed.py::test_full_c4_grid_across_processes runs
    the same grid over the peer-to-peer and the shared-memory transport)."""
    import tempfile
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from stencilflow_amd.lowering import lower
    shape, stages, world = C4_SHAPE, C4_STAGES, 8
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(programs.jacobi3d(shape, stages), os.path.join(tmp, "c4.json"))
        sfir = lower(sf.KernelChainGraph(path))
    exch = LocalExchanger(world)
    views = [exch.for_rank(r) for r in range(world)]
    for v in views:
        v.reserved_cus = 32
    runners = [SlabRunner(sfir, shape, r, world, exchanger=views[r]) for r in range(world)]
    # (four launches per exchange; a launch is three operators since round 5 -- two before: halos of 8 planes)
    assert runners[0].is_chain and runners[0].halo == 12 and runners[3].n_local == 512
    assert "[dense T=3" in runners[0].plan.describe()
    for r in runners:
        r.upload([c4_input(r.lo, r.hi)])
    run_lockstep(runners)
    got = np.empty(shape, np.float32)
    for r in runners:
        part = np.empty(r.local_shape, np.float32)
        r.download([part])
        got[r.lo:r.hi] = part
        r.close()
    assert np.array_equal(got, c4_oracle())


def test_degenerate_programs(tmp_path):
    """tests/degenerate_programs.py on the GPU against the oracle."""
    from tests.degenerate_programs import VALID
    for name, prog in VALID.items():
        path = _write(tmp_path, prog, name)
        ins = _inputs_of(prog)
        want = npo.run_reference(prog, inputs=ins)["b"]
        got, _ = _run_gpu(path, ins)
        assert got["b"].dtype == want.dtype and np.array_equal(got["b"], want), name


def test_command_line_with_a_named_reference_checker(programs_dir, tmp_path):
    """bin/synthesize.py -> bin/run_program.py ... -compare-to-reference: the CPU checker is
    named on the command line (-reference-checker module:function); without one the
    flag fails loudly -- the product never computes on the CPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("SF_REFERENCE_CHECKER", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bin", "synthesize.py"), "float32", "3", "1", "12", "16", "32",
                        "1", "1", "1", "-stencil_shape", "box"], cwd=str(tmp_path), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    prog = r.stdout.strip().split(": ")[-1]
    cmd = [sys.executable, os.path.join(root, "bin", "run_program.py"), prog, "hardware", "-compare-to-reference",
           "-log-level", "1"]
    ok = subprocess.run(cmd + ["-reference-checker", "tests.reference_provider:reference_outputs"], cwd=str(tmp_path),
                        capture_output=True, text=True, env=env)
    assert ok.returncode == 0 and "Results verified" in ok.stdout, ok.stdout[-800:] + ok.stderr[-800:]
    name = os.path.splitext(prog)[0].replace(".", "_")
    out = np.fromfile(str(tmp_path / "results" / name / "b2.dat"), np.float32)
    ref = np.fromfile(str(tmp_path / "results" / name / "reference" / "b2.dat"), np.float32)
    assert out.size == 12 * 16 * 32 and np.array_equal(out, ref)
    bad = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, env=env)
    assert bad.returncode != 0 and "RuntimeError" in bad.stderr


# ---------------------------------------------------------------------------------------------------------------
# Round 5: the streaming dense forms take their planes by LDS-DMA, and a plain sum streams in ANY order of its terms
# (codegen.hpp: stream_schedule) -- the generator's crosses and `diffusion` shapes of extent 3 left the generic kernel.
def _synth_case(tmp_path, dtype, dims, extent, shape, bc, stages=2, seed=41):
    full = list(dims) + [0] * (3 - len(dims))
    ext = [extent if d else 0 for d in full]
    prog, _ = programs.synthesize(dtype, stages, 0.0, *full, *ext, stencil_shape=shape)
    for k in prog["program"].values():
        for f in k["boundary_conditions"]:
            k["boundary_conditions"][f] = dict(bc)
    x = np.random.default_rng(SEED + seed).uniform(-1, 1, dims).astype(dtype)
    return prog, x, sf.KernelChainGraph(_write(tmp_path, prog))


def _plan_inputs(plan, prog, x):
    """inputs in the plan's order: the field `a`, the scalars the generator's `diffusion` / `hotspot` shapes declare"""
    if plan.scalar_names:
        plan.set_scalars([float(str(prog["inputs"][n]["data"]).split(":")[-1]) for n in plan.scalar_names])
    return [x for _ in plan.input_names]


@pytest.mark.parametrize("dims,dtype,shape,bc", [
    ((14, 19, 40), "float32", "cross", {"type": "constant", "value": 0}),
    ((9, 26, 72), "float32", "cross", {"type": "constant", "value": 0.5}),     # float literal: double-typed sum
    ((11, 13, 24), "float64", "cross", {"type": "constant", "value": -1}),
    ((16, 21, 136), "float32", "cross", {"type": "shrink"}),
    ((90, 136), "float32", "cross", {"type": "constant", "value": 2}),
    ((70, 72), "float64", "cross", {"type": "constant", "value": 0.25}),
    ((12, 17, 40), "float32", "diffusion", {"type": "constant", "value": 0}),  # centre first, then the crosses' order
    ((10, 15, 24), "float64", "diffusion", {"type": "constant", "value": 0.5}),
    ((80, 264), "float32", "diffusion", {"type": "constant", "value": 0}),
    ((60, 136), "float32", "cross", {"type": "shrink"}),
    ((8, 12, 24), "float32", "diffusion", {"type": "constant", "value": 0.5}),  # double-typed products and sums
])
def test_generator_crosses_of_extent_three_stream_through_the_dense_kernel(tmp_path, dims, dtype, shape, bc):
    """Radius-3 stars ran on the generic kernel until round 4: their text lists the planes out of order (i-3 .. i+3,
    then the 12 in-plane terms).  The streaming form now adds a term to its output plane at the step by which its plane
    AND every earlier term have arrived -- the in-plane terms three steps after their plane, which the LDS ring keeps
    (SF_LAG 3) -- in the order of the text: bit for bit the oracle's results, whatever the sum's type, in 3-D and 2-D."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    prog, x, chain = _synth_case(tmp_path, dtype, dims, 3, shape, bc)
    got = np.zeros(dims, dtype)
    with Plan(lower(chain)) as plan:
        assert "[dense" in plan.describe() and "[point]" not in plan.describe(), plan.describe()
        src = plan.kernel_source(0)
        assert "#define SF_DENSE_STREAM 1" in src and "#define SF_LAG 3" in src and "offen lds" in src
        plan.run(_plan_inputs(plan, prog, x), [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    if bc["type"] == "shrink":
        inner = tuple(slice(6, -6) for _ in dims)
        assert np.array_equal(got[inner], want[inner])
    else:
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("dims,extent,bc", [
    ((13, 11, 520), 2, {"type": "constant", "value": 0.5}),   # one row of tiles, rows cut into three k-tiles
    ((7, 70, 260), 2, {"type": "constant", "value": -2}),     # partial row tiles, a k-tile of four columns
    ((6, 9, 8), 2, {"type": "constant", "value": 1}),         # the whole grid inside one tile's halo
    ((21, 40, 1028), 1, {"type": "constant", "value": 0.5}),  # 27-point boxes, two per launch, rows cut into k-tiles
    ((9, 37, 512), 1, {"type": "constant", "value": -1}),     # ... whole rows of 512 columns
    ((300, 520), 2, {"type": "constant", "value": 3}),        # 2-D: 25 points
    ((64, 1032), 1, {"type": "constant", "value": 0.5}),      # 2-D: 9 points, two per launch
    ((50, 72), 3, {"type": "constant", "value": -1}),         # 2-D: 49 points, seven open rows
    ((5, 6, 8), 1, {"type": "constant", "value": 2}),         # 27-point boxes on a grid smaller than a tile
])
def test_nonzero_boundary_constants_under_lds_dma(tmp_path, dims, extent, bc):
    """A plane requested by LDS-DMA arrives with ZEROS where it reaches beyond the domain (out-of-range lanes of
    `buffer_load ... lds` write zero, tools/micro/lds_dma_probe.hip); a boundary constant other than zero is written over
    them by the tiles that touch the edge, and over whole planes outside the global domain, before anything reads the
    slot (dense3d.h: sf_fix_boundary).  Partial tiles, k-tiles, grids smaller than a halo: all results bit for bit."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    prog, x, chain = _synth_case(tmp_path, "float32", dims, extent, "box", bc, stages=4 if extent == 1 else 2, seed=42)
    got = np.zeros(dims, np.float32)
    with Plan(lower(chain), options={"dense.t2": 2}) as plan:
        assert "[dense" in plan.describe(), plan.describe()
        assert "offen lds" in plan.kernel_source(0)
        plan.run([x], [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("shape,extent,world", [("cross", 3, 2), ("cross", 3, 3), ("box", 3, 2), ("box", 2, 3), ("cross", 2, 2),
                                                ("cross", 2, 3), ("cross", 1, 2), ("cross", 1, 3)])
def test_streaming_dense_launches_under_slab_decomposition(tmp_path, shape, extent, world):
    """The streaming dense forms on in-process slabs: a launch reaches `extent` planes across a slab boundary (the fused
    pair of radius-2 crosses: four; three radius-1 crosses per launch: three), the planes requested by LDS-DMA include ghost planes, and planes outside the GLOBAL
    domain (not the slab) are the ones that hold the boundary constant."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from stencilflow_amd.lowering import lower
    dims = (36, 14, 40)
    prog, x, chain = _synth_case(tmp_path, "float32", dims, extent, shape, {"type": "constant", "value": 0.5}, stages=3, seed=43)
    sfir = lower(chain)
    exch = LocalExchanger(world)
    options = {"dense.t2": 2} if extent == 2 and shape == "cross" else {"dense.t2": 3, "fuse": 3} if extent == 1 else None
    runners = [SlabRunner(sfir, dims, r, world, exchanger=exch.for_rank(r), groups_per_exchange=1, options=options) for r in range(world)]
    assert all("[dense" in r.plan.describe() for r in runners), runners[0].plan.describe()
    if extent == 2 and shape == "cross":
        assert all("#define SF_RS 2\n" in r.plan.kernel_source(0) for r in runners)
    if extent == 1:  # (three operators per launch: the launch reaches three planes across a slab boundary)
        assert all("#define SF_NST 3\n" in r.plan.kernel_source(0) for r in runners)
    for r in runners:
        r.upload([x[r.lo:r.hi]])
    run_lockstep(runners)
    got = np.zeros(dims, np.float32)
    for r in runners:
        part = np.zeros(r.local_shape, np.float32)
        r.download([part])
        got[r.lo:r.hi] = part
        r.close()
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("pins,mid_slots", [({"k1.bx": 128, "k1.by": 6, "k1.rj": 3}, 1), ({"k1.bx": 128, "k1.by": 4, "k1.rj": 4}, 2),
                                             ({"k1.bx": 64, "k1.by": 8, "k1.rj": 2}, 2)])
def test_fused_dense_form_with_one_and_two_slots_between_the_operators(tmp_path, pins, mid_slots):
    """Two 27-point sums per launch: the input planes are requested a whole step ahead into a ring of two slots; the ring
    between the operators has two slots where 160 KB of LDS allow four, and ONE where they allow three (18-row tiles of
    512 columns: what the planner picks at 512^3) -- operator 2 then reads it first and operator 1's plane goes there at
    the very end of the step, behind a second barrier.  Same results either way."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    dims = (19, 45, 512)
    prog, x, chain = _synth_case(tmp_path, "float32", dims, 1, "box", {"type": "constant", "value": 0}, stages=4, seed=44)
    got = np.zeros(dims, np.float32)
    with Plan(lower(chain), options=dict(pins, **{"dense.t2": 2})) as plan:
        src = plan.kernel_source(0)
        assert "sf_dense3d_f32_t2_" in plan.describe() and "#define SF_IN_SLOTS 2\n" in src, plan.describe()
        assert ("#define SF_MID_SLOTS 1\n" in src) == (mid_slots == 1)
        plan.run([x], [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dims,bc,stages,pins", [
    ((14, 37, 72), {"type": "constant", "value": 0}, 2, {}),
    ((9, 30, 136), {"type": "constant", "value": 0.5}, 4, {}),                               # float literal: double-typed sums
    ((20, 33, 520), {"type": "constant", "value": -1}, 2, {}),                               # rows cut into five k-tiles
    ((7, 5, 8), {"type": "constant", "value": 2}, 2, {}),                                    # the whole grid inside a halo
    ((31, 70, 264), {"type": "constant", "value": 0.25}, 3, {}),                             # the odd one out: wide-star kernel
    ((12, 64, 512), {"type": "constant", "value": 0}, 2, {"k1.bx": 32, "k1.by": 16, "k1.rj": 2}),  # whole waves, rows of 128 columns
    ((12, 64, 512), {"type": "constant", "value": 1}, 2, {"k1.bx": 34, "k1.by": 30, "k1.rj": 1}),  # 1020 threads, one row each
    ((70, 41, 136), {"type": "constant", "value": 0.5}, 4, {"k1.bx": 34, "k1.by": 7, "k1.rj": 2}),  # several chunks of planes
])
def test_radius_two_crosses_two_per_streaming_dense_launch(tmp_path, dims, bc, stages, pins):
    """Round 5: the generator's radius-2 crosses (bin/synthesize.py:19-31,95-101: i-2 .. i+2 first, then the j and k terms)
    two per launch in the dense kernel's fused streaming form -- every operator reaches two planes and rows, its in-plane
    terms join their output plane two steps after their own plane arrived (the input ring and the ring between the
    operators keep two more planes each), tiles overlap by two rows and four columns, blocks of 34-thread rows end in a
    wave with lanes off.  dense.t2=2 forces the form onto grids it would not choose; bit for bit the oracle's results."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    prog, x, chain = _synth_case(tmp_path, "float32", dims, 2, "cross", bc, stages=stages, seed=46)
    got = np.zeros(dims, np.float32)
    with Plan(lower(chain), options=dict(pins, **{"dense.t2": 2})) as plan:
        src = plan.kernel_source(0)
        assert "sf_dense3d_f32_t2_" in plan.describe() and "[point]" not in plan.describe(), plan.describe()
        assert "#define SF_RS 2\n" in src and "#define SF_LAG 2\n" in src and "#define SF_LAG2 2\n" in src and "offen lds" in src
        if pins:
            assert "block %dx%d rows/thread %d" % (pins["k1.bx"], pins["k1.by"], pins["k1.rj"]) in plan.describe(), plan.describe()
        plan.run([x], [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("shape,dims,bc,stages,pins", [
    ("cross", (14, 37, 72), {"type": "constant", "value": 0}, 3, {}),
    ("cross", (9, 30, 136), {"type": "constant", "value": 0.5}, 6, {}),                         # float literal: double-typed sums
    ("cross", (20, 33, 520), {"type": "constant", "value": -1}, 4, {}),                         # five k-tiles; the fourth operator alone
    ("cross", (7, 5, 8), {"type": "constant", "value": 2}, 3, {}),                              # the whole grid inside a halo
    ("box", (12, 40, 264), {"type": "constant", "value": 0.25}, 3, {}),                         # three 27-point sums
    ("box", (33, 64, 512), {"type": "constant", "value": 1}, 5, {}),                            # three and two
    ("cross", (12, 64, 512), {"type": "constant", "value": 0}, 3, {"k1.bx": 34, "k1.by": 30, "k1.rj": 1}),  # 1020 threads
    ("cross", (70, 41, 136), {"type": "constant", "value": 0.5}, 6, {"k1.bx": 32, "k1.by": 8, "k1.rj": 2}),  # chunks of planes
])
def test_three_radius_one_sums_per_streaming_dense_launch(tmp_path, shape, dims, bc, stages, pins):
    """dense.t2=3, fuse=3: three radius-1 plain sums per launch of the dense kernel's fused streaming form -- the generator's
    crosses of extent 1 (the benchmark's operator: i-1, i+1, then the in-plane terms, which join their plane a step late) and
    its 27-point boxes; the third operator reads a second LDS ring, tiles overlap by two rows, a row cut into tiles has
    no halo columns in LDS.  Bit for bit the oracle's results."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    prog, x, chain = _synth_case(tmp_path, "float32", dims, 1, shape, bc, stages=stages, seed=47)
    got = np.zeros(dims, np.float32)
    with Plan(lower(chain), options=dict(pins, **{"dense.t2": 3, "fuse": 3})) as plan:
        src = plan.kernel_source(0)
        assert "sf_dense3d_f32_t3_" in plan.describe() and "[point]" not in plan.describe(), plan.describe()
        assert "#define SF_NST 3\n" in src and "struct sf_dense3 {" in src and "offen lds" in src
        assert ("#define SF_LAG 1\n" in src) == (shape == "cross")
        plan.run([x], [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("extent,dims,dtype,bc,stages,options", [
    (1, (14, 37, 72), "float32", {"type": "constant", "value": 0}, 3, {"dense.t2": 3}),
    (1, (9, 30, 136), "float32", {"type": "constant", "value": 0.5}, 6, {"dense.t2": 3}),        # float literal: double-typed sums of float products
    (1, (20, 33, 520), "float32", {"type": "constant", "value": -1}, 4, {"dense.t2": 3}),
    (2, (14, 37, 72), "float32", {"type": "constant", "value": 0}, 2, {"dense.t2": 2}),
    (2, (9, 30, 136), "float32", {"type": "constant", "value": 0.5}, 4, {"dense.t2": 2}),
    (2, (31, 70, 264), "float32", {"type": "constant", "value": 0.25}, 3, {"dense.t2": 2}),       # the odd one out: wide-star kernel
])
def test_weighted_star_sums_in_the_fused_streaming_forms(tmp_path, extent, dims, dtype, bc, stages, options):
    """The generator's `diffusion` shapes (bin/synthesize.py: every term of the cross times a scalar of its own, the centre
    first) in the dense kernel's fused streaming forms: a product is formed in the common type of its factor and its
    operand and then converted to the sum's, term by term in the order of the text; three per launch at extent 1, two at
    extent 2.  512^3 float32 (profiles/r05_diffusion_fused.log): 98.7 -> 85.4 us per operator at extent 1 (star kernel
    before), 182.8 -> 144.7 at extent 2 (wide-star kernel before).  Bit for bit the oracle's results."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    prog, x, chain = _synth_case(tmp_path, dtype, dims, extent, "diffusion", bc, stages=stages, seed=48)
    got = np.zeros(dims, dtype)
    with Plan(lower(chain), options=options) as plan:
        src = plan.kernel_source(0)
        assert ("sf_dense3d_f32_t3_" if extent == 1 else "sf_dense3d_f32_t2_") in plan.describe() and "[point]" not in plan.describe(), plan.describe()
        assert "offen lds" in src and " * (float)g" in src  # (a factor per term)
        plan.run(_plan_inputs(plan, prog, x), [got], 1)
    want = npo.run_reference(prog, inputs={"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want, equal_nan=True)


def test_full_size_radius_three_cross_bit_exact():
    """The generator's radius-3 cross at the benchmark's 512^3 (bench.py's `cross3` workload): all 134 million results
    of a two-operator chain against the C oracle -- the tile, chunk and ring shapes the planner picks at full size."""
    import tempfile
    from oracle import c_oracle
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    n = 512
    prog, _ = programs.synthesize("float32", 2, 0.0, n, n, n, 3, 3, 3)
    x = np.random.default_rng(SEED + 45).uniform(-1, 1, (n, n, n)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        chain = sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "w.json")))
    got = np.zeros((n, n, n), np.float32)
    with Plan(lower(chain)) as plan:
        assert "[dense" in plan.describe() and "block 64x4 rows/thread 2" in plan.describe(), plan.describe()
        plan.run([x], [got], 1)
    ref = c_oracle.CompiledReference(prog)
    ref.threads = _oracle_threads()
    want = ref.run({"a": x})[prog["outputs"][0]]
    assert np.array_equal(got, want), npo.max_rel_err(want, got)
