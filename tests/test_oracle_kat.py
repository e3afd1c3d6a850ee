"""Pins the oracle: closed-form known answers implied by the reference's own
test programs, the one vector the reference's Simulator yields,
and agreement of the NumPy and the C restatement."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle, numpy_oracle as npo


def _run(programs_dir, name, **kw):
    return npo.run_reference(os.path.join(programs_dir, name + ".json"),
                             input_directory=programs_dir, **kw)


def test_jacobi2d_constant_input(programs_dir):
    # a == 1, BC 0: b = 0.25 * (#in-domain neighbours); exact in f32
    b = _run(programs_dir, "jacobi2d_128x128")["b"]
    assert b.dtype == np.float32 and b.shape == (128, 128)
    expect = np.full((128, 128), 1.0, np.float32)
    expect[0, :] = expect[-1, :] = expect[:, 0] = expect[:, -1] = 0.75
    for c in [(0, 0), (0, -1), (-1, 0), (-1, -1)]:
        expect[c] = 0.5
    assert np.array_equal(b, expect)


def test_jacobi3d_zero_input_bc_one(programs_dir):
    # a == 0 (data/zeros_32x32x32_fp32.dat), BC 1.0:
    # b = 0.16666666 * (#out-of-domain neighbours), f64 product rounded to f32
    b = _run(programs_dir, "jacobi3d_32x32x32")["b"]
    n = 32
    idx = np.arange(n)
    edge = ((idx == 0) | (idx == n - 1)).astype(np.int64)
    count = edge[:, None, None] + edge[None, :, None] + edge[None, None, :]
    expect = (0.16666666 * count.astype(np.float64)).astype(np.float32)
    assert np.array_equal(b, expect)
    assert b[5, 5, 5] == 0 and b[0, 0, 0] == np.float32(0.16666666 * 3)


def test_c1_first_stage_and_invariants(programs_dir):
    r = _run(programs_dir, "jacobi3d_32x32x32_8itr_8vec", return_all=True)
    b0, b7 = r["b0"], r["b7"]
    n = 32
    idx = np.arange(n)
    inside = 6 - (((idx == 0) | (idx == n - 1)).astype(np.int64)[:, None, None]
                  + ((idx == 0) | (idx == n - 1)).astype(np.int64)[None, :, None]
                  + ((idx == 0) | (idx == n - 1)).astype(np.int64)[None, None, :])
    assert np.array_equal(
        b0, (0.16666666 * inside.astype(np.float64)).astype(np.float32))
    # deeper stages: no closed form -> symmetry and range invariants
    assert np.array_equal(b7, b7[::-1, :, :])
    assert np.array_equal(b7, b7[:, ::-1, :])
    assert np.array_equal(b7, b7[:, :, ::-1])
    assert np.array_equal(b7, b7.transpose(1, 0, 2))
    assert np.array_equal(b7, b7.transpose(0, 2, 1))
    assert 0.0 <= b7.min() and b7.max() <= 1.0
    assert b7[16, 16, 16] == b7.max() and b7[0, 0, 0] == b7.min()
    # the "_8vec" file is the "_4vec" one (vectorization 4) and W never
    # changes results (sdfg_generator.py:594-595)
    r4 = _run(programs_dir, "jacobi3d_32x32x32_8itr")
    assert np.array_equal(r4["b7"], b7)


@pytest.mark.parametrize("name,expected", [
    ("simulator", [[5.14, 4.14, 5.14], [11.14, 7.14, 8.14]]),
    ("simulator2", [[3, 4, 3], [4, 5, 4], [3, 4, 3]]),
    ("simulator4", [[3, 3, 3], [3, 3, 3], [3, 3, 3]]),
])
def test_small_programs_closed_form(programs_dir, name, expected):
    res = _run(programs_dir, name)["res"]
    assert res.dtype == np.float64
    assert np.allclose(res, np.array(expected, dtype=np.float64), rtol=0,
                       atol=1e-12)


def test_simulator9_is_affine_in_input(programs_dir):
    with open(os.path.join(programs_dir, "simulator9.json")) as f:
        prog = json.load(f)
    arr = np.array(prog["inputs"]["arrA"]["data"], dtype=np.float64)
    res = _run(programs_dir, "simulator9")["res"]
    assert np.array_equal(res.ravel(), 2 * arr.ravel() + 3)


def test_simulator12_matches_reference_simulator(programs_dir, golden_dir):
    with open(os.path.join(golden_dir, "simulator12_expected.json")) as f:
        fixture = json.load(f)
    res = _run(programs_dir, "simulator12")["res"]
    assert np.array_equal(res.ravel(), np.array(fixture["result"]["res"]))


def test_varying_dimensionality(programs_dir):
    out = _run(programs_dir, "varying_dimensionality")["out"]
    assert out.dtype == np.float32 and out.shape == (8, 16, 32)
    f32, f64 = np.float32, np.float64
    i, j, k = np.meshgrid(np.arange(8), np.arange(16), np.arange(32),
                          indexing="ij")
    in2d_next = np.where(i >= 7, f64(1.0), f64(f32(0.3)))
    in3d_next = np.where((i >= 7) | (j >= 15) | (k >= 31), 1.0, 0.4)
    # source order, each partial sum in the C++ type of its operands
    acc = f64(0.1) + f64(f32(0.2))          # double + float
    acc = acc + 1.0                          # in1d[k+42]: always out of domain
    acc = acc + f64(f32(0.3))
    acc = acc + in2d_next
    acc = acc + 0.4
    acc = acc + in3d_next
    assert np.array_equal(out, acc.astype(f32))


def test_far_offsets_and_ternary(programs_dir):
    res = _run(programs_dir, "simulator11")["res"]
    arr = np.arange(9, dtype=np.float64).reshape(3, 3)
    ka = arr + 1.0
    kb = np.zeros((3, 3))
    kb[:, 0] = ka[:, 2]  # kA[j,k+2]; kA[j,k-100] is always the BC 0.0
    assert np.array_equal(res, ka + kb)
    res7 = _run(programs_dir, "simulator7")["res"]
    assert res7.shape == (3, 3) and np.isfinite(res7).all()


def test_c_restatement_is_bit_identical(programs_dir):
    for name in sorted(f[:-5] for f in os.listdir(programs_dir)
                       if f.endswith(".json")):
        path = os.path.join(programs_dir, name + ".json")
        a = npo.run_reference(path, input_directory=programs_dir,
                              return_all=True)
        b = c_oracle.CompiledReference(path).run(
            input_directory=programs_dir, return_all=True)
        for k in a:
            assert a[k].dtype == b[k].dtype, (name, k)
            assert np.array_equal(a[k], b[k]), (name, k)


def test_c_restatement_random_inputs():
    from stencilflow_amd import programs
    rng = np.random.default_rng(20261003)
    p = programs.jacobi3d((20, 12, 16), 5)
    x = rng.uniform(-1, 1, (20, 12, 16)).astype(np.float32)
    a = npo.run_reference(p, {"a": x})["b4"]
    b = c_oracle.CompiledReference(p).run({"a": x})["b4"]
    assert np.array_equal(a, b)
    p = programs.jacobi2d((33, 20), 4)
    x = rng.uniform(-1, 1, (33, 20)).astype(np.float32)
    assert np.array_equal(npo.run_reference(p, {"a": x})["b3"],
                          c_oracle.CompiledReference(p).run({"a": x})["b3"])
    p = programs.diffusion_advection_laplacian((10, 12, 16))
    x = rng.uniform(-1, 1, (10, 12, 16))
    assert np.array_equal(npo.run_reference(p, {"a": x})["lap"],
                          c_oracle.CompiledReference(p).run({"a": x})["lap"])


def test_integer_bc_literal_keeps_float_type():
    """bin/synthesize.py writes BC ``"value": 0`` (an int): the C++ select then
    has type float, so the sum is rounded per add in f32 -- a different result
    from a 0.0 (double) literal.  Both typings must be honoured."""
    from stencilflow_amd import programs
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (6, 8, 8)).astype(np.float32)
    pd = programs.jacobi3d((6, 8, 8), 1, bc_value=0.0)
    pi = programs.jacobi3d((6, 8, 8), 1, bc_value=0)
    rd = npo.run_reference(pd, {"a": x})["b0"]
    ri = npo.run_reference(pi, {"a": x})["b0"]
    xp = np.pad(x, 1).astype(np.float32)
    s32 = ((((xp[:-2, 1:-1, 1:-1] + xp[2:, 1:-1, 1:-1]) + xp[1:-1, :-2, 1:-1])
            + xp[1:-1, 2:, 1:-1]) + xp[1:-1, 1:-1, :-2]) + xp[1:-1, 1:-1, 2:]
    assert s32.dtype == np.float32
    assert np.array_equal(ri, (0.16666666 * s32.astype(np.float64)).astype(
        np.float32))
    xq = xp.astype(np.float64)
    s64 = ((((xq[:-2, 1:-1, 1:-1] + xq[2:, 1:-1, 1:-1]) + xq[1:-1, :-2, 1:-1])
            + xq[1:-1, 2:, 1:-1]) + xq[1:-1, 1:-1, :-2]) + xq[1:-1, 1:-1, 2:]
    assert np.array_equal(rd, (0.16666666 * s64).astype(np.float32))
    assert not np.array_equal(rd, ri)
