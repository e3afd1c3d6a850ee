"""Vectors produced by the REFERENCE itself: its ``Simulator`` run on small
float32 / mixed-dtype programs and on BASELINE.json's configs[0] (jacobi3d 32^3,
8 operators, float32), captured by tests/golden/make_simulator_fixtures.py into
tests/golden/simulator_vectors.json (+ raw ``c1_*.dat`` files).

What they pin.  The Simulator evaluates an operator as
``data_type(eval_expr(var_map, computation))`` on NumPy scalars and Python
floats (reference stencilflow/kernel.py:700-709), i.e. with NumPy's typing of
literals, while the reference's CPU program is DaCe-generated C++.  So

* the oracle evaluated with NumPy's literal typing (``typing="nep50"``) must
  reproduce every vector BIT FOR BIT -- this pins indexing, boundary selection,
  operand order, chain mechanics and dtype casts of the float32 path on data
  that come from the reference;
* the oracle under its own contract (C++ typing: DESIGN.md §2) and the HIP
  backend must agree with the vectors PER POINT to within the tolerance
  BASELINE.json's north_star states, 1e-6 relative (round 3; rounds 1-2 scaled by
  the field's maximum): relative to the point's own value wherever that is
  meaningful, and -- for the float32 programs on signed random data listed in
  CANCELLING, whose sums cancel to ~0 at some points (a float32 sum of O(1)
  operands that comes out at 1e-3 cannot be right to 1e-6 of ITSELF under a
  different rounding of that sum) -- relative to the largest magnitude among the
  point and its nearest neighbours, where the last operator's operands live;
* programs whose arithmetic is exact in float32 (``*_exact``, the fork/join and
  the float64 program) must agree bit for bit under both typings.
"""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle, numpy_oracle as npo

TOL = 1e-6  # BASELINE.json north_star: "within 1e-6 relative for float32"
BIT_EXACT_UNDER_BOTH_TYPINGS = {"f32_jacobi7_exact", "mixed_to_f64", "f32_box_exact", "f32_fork_join"}
# signed random float32 data: some results cancel to ~0 (per-point error relative to the result itself up to 2.7e-5,
# relative to the operands' magnitude at most 2.8e-7)
CANCELLING = {"f32_chain2", "f32_chain8", "f32_jacobi7", "f32_weighted_bc", "f32_hotspot2", "f32_wide_cross2"}


def _vectors(golden_dir):
    with open(os.path.join(golden_dir, "simulator_vectors.json")) as f:
        return json.load(f)


def _names():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simulator_vectors.json")
    with open(here) as f:
        return sorted(json.load(f)["programs"])


def _expected(entry):
    dims = entry["program"]["dimensions"]
    return {k: np.array(v["values"], dtype=v["dtype"]).reshape(dims) for k, v in entry["result"].items()}


def _within_tolerance(expected, got, name=None):
    """Per point: |difference| <= TOL x the point's own magnitude; for the named cancelling programs
    TOL x the largest magnitude among the point and its nearest neighbours."""
    from scipy.ndimage import maximum_filter
    e, g = expected.astype(np.float64), got.astype(np.float64)
    if name is not None and name not in CANCELLING:
        return npo.max_rel_err(expected, got) <= TOL
    scale = maximum_filter(np.maximum(np.abs(e), np.abs(g)), size=3, mode="nearest")
    return bool(np.all(np.abs(e - g) <= TOL * scale))


def test_fixture_inventory(golden_dir):
    v = _vectors(golden_dir)
    assert "Simulator" in v["source"]
    assert len(v["programs"]) >= 9 and "f32_chain8" in v["programs"]
    assert set(v["large"]) == {"c1_file_input", "c1_random_input"}


@pytest.mark.parametrize("name", _names())
def test_oracle_reproduces_the_reference_simulator(golden_dir, name):
    entry = _vectors(golden_dir)["programs"][name]
    expected = _expected(entry)
    weak = npo.run_reference(entry["program"], typing="nep50")
    own = npo.run_reference(entry["program"])
    compiled = c_oracle.CompiledReference(entry["program"]).run()
    for out, exp in expected.items():
        assert weak[out].dtype == exp.dtype
        assert np.array_equal(weak[out], exp), (name, out)  # bit for bit
        assert np.array_equal(own[out], compiled[out]), (name, out)  # NumPy == C restatement
        if name in BIT_EXACT_UNDER_BOTH_TYPINGS:
            assert np.array_equal(own[out], exp), (name, out)
        else:
            assert _within_tolerance(exp, own[out], name), (name, out)


def _c1(golden_dir, programs_dir, tag):
    info = _vectors(golden_dir)["large"][tag]
    expected = np.fromfile(os.path.join(golden_dir, info["output"]), np.float32).reshape(info["shape"])
    inputs = None
    if info["input"]:
        inputs = {"a": np.fromfile(os.path.join(golden_dir, info["input"]), np.float32).reshape(info["shape"])}
    path = os.path.join(programs_dir, "jacobi3d_32x32x32_8itr_8vec.json")
    return path, inputs, expected


@pytest.mark.parametrize("tag", ["c1_file_input", "c1_random_input"])
def test_baseline_config0_against_the_reference_simulator(golden_dir, programs_dir, tag):
    """BASELINE.json configs[0]: the reference's own jacobi3d_32x32x32_8itr_8vec.json."""
    path, inputs, expected = _c1(golden_dir, programs_dir, tag)
    weak = npo.run_reference(path, inputs=inputs, input_directory=programs_dir, typing="nep50")["b7"]
    assert np.array_equal(weak, expected)  # bit for bit, 32768 values, 8 operators deep
    own = npo.run_reference(path, inputs=inputs, input_directory=programs_dir)["b7"]
    assert _within_tolerance(expected, own)
    # at this depth the two typings also agree point by point (tests/test_rounding_envelope.py)
    assert npo.max_rel_err(expected, own) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_hip_against_the_reference_simulator(golden_dir, tmp_path, name):
    from tests.test_gpu_parity import _inputs_of, _run_gpu
    entry = _vectors(golden_dir)["programs"][name]
    path = str(tmp_path / (name + ".json"))
    with open(path, "w") as f:
        json.dump(entry["program"], f)
    ins = _inputs_of(path)
    got, _ = _run_gpu(path, ins)
    own = npo.run_reference(path, inputs=ins)
    for out, exp in _expected(entry).items():
        assert got[out].dtype == exp.dtype
        assert np.array_equal(got[out], own[out]), (name, out)  # HIP == oracle, bit for bit
        if name in BIT_EXACT_UNDER_BOTH_TYPINGS:
            assert np.array_equal(got[out], exp), (name, out)  # HIP == reference, bit for bit
        else:
            assert _within_tolerance(exp, got[out], name), (name, out)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["c1_file_input", "c1_random_input"])
@pytest.mark.parametrize("options", [None, {"fuse": 1}, {"generic_only": 1}])
def test_hip_baseline_config0_against_the_reference_simulator(golden_dir, programs_dir, tag, options):
    from tests.test_gpu_parity import _inputs_of, _run_gpu
    path, inputs, expected = _c1(golden_dir, programs_dir, tag)
    ins = inputs or _inputs_of(path, programs_dir)
    got, _ = _run_gpu(path, ins, options=options)
    assert _within_tolerance(expected, got["b7"])
    assert npo.max_rel_err(expected, got["b7"]) <= TOL


@pytest.mark.parametrize("name", ["f64_chain3", "f32_hotspot2"])
def test_oracle_reproduces_the_reference_simulator_on_chains(golden_dir, name):
    """Two chains evaluated by the reference's own Simulator (tests/golden/make_simulator_chain_fixtures.py):
    the structure of BASELINE.json's configs[3] (C5: diffusion -> advection -> laplacian, float64) and two
    float32 operators of the generator's hotspot shape sharing an auxiliary field.  The NumPy oracle with
    the Simulator's typing reproduces both bit for bit; under the contract's typing the float64 chain is
    bit-exact as well (in float64 the typings cannot differ) and the float32 one within 1e-6; the C
    restatement equals the NumPy one."""
    with open(os.path.join(golden_dir, "simulator_chains.json")) as f:
        doc = json.load(f)
    assert "Simulator" in doc["source"]
    entry = doc["programs"][name]
    (out, expected), = _expected(entry).items()
    assert float(np.abs(expected).max()) > 0.0
    assert np.array_equal(npo.run_reference(entry["program"], typing="nep50")[out], expected)
    own = npo.run_reference(entry["program"])[out]
    assert np.array_equal(c_oracle.CompiledReference(entry["program"]).run()[out], own)
    if expected.dtype == np.float64:
        assert np.array_equal(own, expected)
    else:
        assert _within_tolerance(expected, own, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f64_chain3", "f32_hotspot2"])
@pytest.mark.parametrize("options", [None, {"fuse": 1}, {"generic_only": 1}])
def test_hip_against_the_reference_simulator_on_chains(golden_dir, tmp_path, name, options):
    """The HIP path on the two chain vectors of the reference's Simulator: equal to the oracle bit for
    bit, equal to the reference bit for bit in float64 and within 1e-6 in float32 -- fused (the float32
    pair takes the auxiliary rows of its first operator over to the second), operator by operator, and
    on the generic kernel."""
    from tests.test_gpu_parity import _inputs_of, _run_gpu
    with open(os.path.join(golden_dir, "simulator_chains.json")) as f:
        entry = json.load(f)["programs"][name]
    path = str(tmp_path / (name + ".json"))
    with open(path, "w") as f:
        json.dump(entry["program"], f)
    ins = _inputs_of(path)
    got, _ = _run_gpu(path, ins, options=options)
    own = npo.run_reference(path, inputs=ins)
    (out, exp), = _expected(entry).items()
    assert got[out].dtype == exp.dtype
    assert np.array_equal(got[out], own[out])
    if exp.dtype == np.float64:
        assert np.array_equal(got[out], exp)
    else:
        assert _within_tolerance(exp, got[out], name)


# ---- radius-2 stars (tests/golden/simulator_wide.json, round 3): what pins kernels/wstar3d.h
# against the reference itself -----------------------------------------------------------------
WIDE = ["f32_wide_cross_exact", "f32_wide_cross2", "f64_wide_diffusion"]
WIDE_BIT_EXACT = {"f32_wide_cross_exact", "f64_wide_diffusion"}


def _wide(golden_dir, name):
    with open(os.path.join(golden_dir, "simulator_wide.json")) as f:
        doc = json.load(f)
    assert "Simulator" in doc["source"]
    return doc["programs"][name]


@pytest.mark.parametrize("name", WIDE)
def test_oracle_reproduces_the_reference_simulator_on_radius_2_stars(golden_dir, name):
    """The operators the reference's generator emits for an extent of 2 (bin/synthesize.py:19-31),
    evaluated by the reference's Simulator: bit for bit under its typing; under the contract's
    bit for bit where float32 arithmetic is exact and in float64, per point within 1e-6 otherwise;
    NumPy and C restatements equal."""
    entry = _wide(golden_dir, name)
    (out, exp), = _expected(entry).items()
    assert np.array_equal(npo.run_reference(entry["program"], typing="nep50")[out], exp)
    own = npo.run_reference(entry["program"])[out]
    assert np.array_equal(c_oracle.CompiledReference(entry["program"]).run()[out], own)
    if name in WIDE_BIT_EXACT:
        assert np.array_equal(own, exp)
    else:
        assert _within_tolerance(exp, own, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", WIDE)
@pytest.mark.parametrize("options", [None, {"fuse": 1}, {"generic_only": 1}, {"k1.bx": 64, "k1.by": 2, "k1.rj": 4, "dense.t2": 0}])
def test_hip_against_the_reference_simulator_on_radius_2_stars(golden_dir, tmp_path, name, options):
    """The HIP path on the same vectors: the fused wide-star kernel (two operators per launch for
    the chain), one operator per launch, the generic kernel, and a pinned tile with thread rows
    that exchange two rows through LDS -- equal to the oracle bit for bit, to the reference bit for
    bit where arithmetic is exact / float64 and per point within 1e-6 otherwise."""
    from tests.test_gpu_parity import _inputs_of, _run_gpu
    entry = _wide(golden_dir, name)
    path = str(tmp_path / (name + ".json"))
    with open(path, "w") as f:
        json.dump(entry["program"], f)
    ins = _inputs_of(path)
    got, plan_text = _run_gpu(path, ins, options=options)
    if not (options or {}).get("generic_only"):
        assert "[wide star" in plan_text
    own = npo.run_reference(path, inputs=ins)
    (out, exp), = _expected(entry).items()
    assert got[out].dtype == exp.dtype
    assert np.array_equal(got[out], own[out])
    if name in WIDE_BIT_EXACT:
        assert np.array_equal(got[out], exp)
    else:
        assert _within_tolerance(exp, got[out], name)


# ---- the six CANCELLING programs again on data in [0, 1) (tests/golden/simulator_unit.json, round 4): no sum cancels,
# ---- so the rule of BASELINE.json's north_star applies as it stands -- per point, 1e-6 of the point's own value
def _unit_names():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simulator_unit.json")
    with open(here) as f:
        return sorted(json.load(f)["programs"])


def _unit(golden_dir, name):
    with open(os.path.join(golden_dir, "simulator_unit.json")) as f:
        return json.load(f)["programs"][name]


def test_unit_twins_cover_every_cancelling_program():
    assert set(_unit_names()) == {n + "_unit" for n in CANCELLING}


@pytest.mark.parametrize("name", _unit_names())
def test_oracle_against_the_reference_simulator_per_point_on_unit_data(golden_dir, name):
    entry = _unit(golden_dir, name)
    weak = npo.run_reference(entry["program"], typing="nep50")
    own = npo.run_reference(entry["program"])
    compiled = c_oracle.CompiledReference(entry["program"]).run()
    for out, exp in _expected(entry).items():
        assert float(exp.min()) > 0.0  # nothing cancels: the strict rule is meaningful at every point
        assert np.array_equal(weak[out], exp), (name, out)  # the Simulator's own typing: bit for bit
        assert np.array_equal(own[out], compiled[out]), (name, out)
        assert npo.max_rel_err(exp, own[out]) <= TOL, (name, out, npo.max_rel_err(exp, own[out]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", _unit_names())
def test_hip_against_the_reference_simulator_per_point_on_unit_data(golden_dir, tmp_path, name):
    from tests.test_gpu_parity import _inputs_of, _run_gpu
    entry = _unit(golden_dir, name)
    path = str(tmp_path / (name + ".json"))
    with open(path, "w") as f:
        json.dump(entry["program"], f)
    ins = _inputs_of(path)
    got, _ = _run_gpu(path, ins)
    own = npo.run_reference(path, inputs=ins)
    for out, exp in _expected(entry).items():
        assert np.array_equal(got[out], own[out]), (name, out)  # HIP == oracle, bit for bit
        assert npo.max_rel_err(exp, got[out]) <= TOL, (name, out)  # HIP vs the reference: 1e-6 per point
