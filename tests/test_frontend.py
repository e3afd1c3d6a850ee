"""Host-side logic: program parsing, the KernelChainGraph facade against
structure fixtures captured from the reference, helper functions (mirroring
the reference's HelperTest, test/test_stencilflow.py:114-162), error classes."""
import json
import os

import numpy as np
import pytest

import stencilflow_amd as sf
from stencilflow_amd import helper
from stencilflow_amd.lowering import lower


def _norm(ix):
    return tuple(-99 if v is None else v for v in ix)


def test_structure_matches_reference(programs_dir, golden_dir):
    with open(os.path.join(golden_dir, "reference_structure.json")) as f:
        ref = json.load(f)
    assert len(ref) == 19
    for name, want in ref.items():
        chain = sf.KernelChainGraph(os.path.join(programs_dir, name + ".json"))
        assert chain.dimensions == want["dimensions"]
        assert chain.kernel_dimensions == want["kernel_dimensions"]
        assert chain.vectorization == want["vectorization"]
        assert list(chain.outputs) == want["outputs"]
        edges = sorted([type(u).__name__, u.name, type(v).__name__, v.name]
                       for u, v in chain.graph.edges())
        assert edges == want["edges"], name
        for kname, kw in want["kernels"].items():
            k = chain.kernel_nodes[kname]
            assert repr(k.data_type) == kw["data_type"]
            assert k.kernel_string == kw["kernel_string"]
            got = {f: sorted(map(_norm, lst))
                   for f, lst in k.graph.accesses.items()}
            exp = {f: sorted(set(map(_norm, lst)))
                   for f, lst in kw["accesses"].items()}
            assert got == exp, (name, kname)
            assert sorted(k.inputs) == kw["reads"]
            for f, dims in kw["input_dims"].items():
                assert k.inputs[f]["input_dims"] == dims
        for iname, iw in want["inputs"].items():
            assert chain.inputs[iname]["input_dims"] == iw["input_dims"]
            assert repr(chain.input_nodes[iname].data_type) == iw["data_type"]
        assert [k.name for k in chain.topological_kernels()
                ] == want["topological_kernels"]
        assert chain.minimum_communication_volume(
        ) == want["minimum_communication_volume"]
        assert {k: list(v) for k, v in chain.operation_count().items()
                } == want["operation_count"]


def test_helper_functions(programs_dir, tmp_path, monkeypatch):
    assert helper.max_dict_entry_key({"a": [1, 0, 0], "b": [0, 1, 0],
                                      "c": [0, 0, 1]}) == "a"
    assert helper.list_add_cwise([1, 2, 3], [3, 2, 1]) == [4, 4, 4]
    assert helper.list_subtract_cwise([1, 2, 3], [1, 2, 3]) == [0, 0, 0]
    assert helper.dim_to_abs_val([3, 2, 1], [10, 10, 10]) == 321
    assert helper.convert_3d_to_1d(dimensions=[10, 10, 10],
                                   index=[3, 2, 1]) == 321
    f64 = sf.str_to_dtype("float64")
    for ext in ("csv", "dat"):
        arr = helper.load_array({
            "data": os.path.join(programs_dir, "helper_test." + ext),
            "data_type": f64})
        assert list(arr) == [7.0, 7.0]
    monkeypatch.chdir(tmp_path)
    out = np.array([1.0, 2.0, 3.0])
    helper.save_array(out, "test.dat")
    back = helper.load_array({"data": "test.dat", "data_type": f64})
    assert helper.arrays_are_equal(out, back)
    assert sorted(helper.unique([1.0, 2.0, 1.0])) == [1.0, 2.0]
    a = helper.aligned(np.arange(7, dtype=np.float32)[1:], 64)
    assert a.ctypes.data % 64 == 0 and list(a) == [1, 2, 3, 4, 5, 6]
    c = helper.load_array({"data": "constant:0.5", "data_type": f64},
                          shape=[2, 3])
    assert c.shape == (2, 3) and (c == 0.5).all()
    assert helper.load_array({"data": "constant:2", "data_type": f64,
                              "input_dims": []}) == 2.0


def test_comparison_rules():
    a = np.array([1.0, 2.0, 3.0])
    assert sf.arrays_are_equal(a, a * (1 + 5e-6))
    assert not sf.arrays_are_equal(a, a * (1 + 5e-5))
    # the reference rule lets negative data pass trivially; the strict one not
    assert sf.arrays_are_equal(-a, -2 * a)
    assert not sf.arrays_match(-a, -2 * a)
    assert sf.arrays_match(a, a * (1 + 5e-7)) and not sf.arrays_match(
        a, a * (1 + 5e-6))
    assert not sf.arrays_match(np.array([np.nan]), np.array([np.nan]))


def test_error_classes(tmp_path, programs_dir):
    with pytest.raises(RuntimeError):
        sf.parse_json(str(tmp_path / "missing.json"))
    with pytest.raises(AttributeError):
        sf.str_to_dtype("float13")
    with open(os.path.join(programs_dir, "jacobi2d_128x128.json")) as f:
        prog = json.load(f)
    cyc = json.loads(json.dumps(prog))
    cyc["program"]["c"] = {
        "computation_string": "c = b[j,k] + d[j,k]",
        "boundary_conditions": {}, "data_type": "float32"}
    cyc["program"]["d"] = {
        "computation_string": "d = c[j,k]",
        "boundary_conditions": {}, "data_type": "float32"}
    cyc["outputs"] = ["d"]
    p = tmp_path / "cyc.json"
    p.write_text(json.dumps(cyc))
    with pytest.raises(ValueError, match="Cycle detected"):
        sf.KernelChainGraph(str(p))
    bad = json.loads(json.dumps(prog))
    bad["program"]["b"]["boundary_conditions"]["a"]["type"] = "periodic"
    p = tmp_path / "bad.json"
    p.write_text(json.dumps(bad))
    with pytest.raises(ValueError, match="Unsupported boundary condition"):
        sf.KernelChainGraph(str(p))
    vec = json.loads(json.dumps(prog))
    vec["dimensions"] = [128, 130]
    vec["vectorization"] = 4
    p = tmp_path / "vec.json"
    p.write_text(json.dumps(vec))
    with pytest.raises(ValueError, match="vectorization"):
        lower(sf.KernelChainGraph(str(p)))
    orphan = json.loads(json.dumps(prog))
    orphan["program"]["z"] = {
        "computation_string": "z = a[j,k]",
        "boundary_conditions": {}, "data_type": "float32"}
    p = tmp_path / "orphan.json"
    p.write_text(json.dumps(orphan))
    with pytest.raises(ValueError, match="Orphan"):
        lower(sf.KernelChainGraph(str(p)))


def test_typing_of_expressions(programs_dir):
    chain = sf.KernelChainGraph(
        os.path.join(programs_dir, "varying_dimensionality.json"))
    text = lower(chain)
    assert "acc in1d_42 in1d f64 constant 1.0 x x 42" in text
    assert "acc in2d_0_0 in2d f32 none - 0 x 0" in text
    assert "let double out = " in text
    chain = sf.KernelChainGraph(
        os.path.join(programs_dir, "jacobi2d_128x128.json"))
    text = lower(chain)
    assert "dims 2 128 128" in text and "acc a_m1_0 a f64 constant 0.0 -1 0" in text


def test_thousand_stage_chain_is_linear_time(tmp_path):
    import time
    from stencilflow_amd import programs
    path = programs.write_program(programs.jacobi3d((512, 512, 512), 1000),
                                  str(tmp_path / "c3.json"))
    t = time.time()
    chain = sf.KernelChainGraph(path)
    text = lower(chain)
    assert time.time() - t < 5.0  # the reference needs ~41 s
    assert len(chain.kernel_nodes) == 1000 and text.count("\nkernel ") == 1000
    assert chain.cell_updates() == 512**3 * 1000
    assert chain.algorithmic_bytes() == 512**3 * 1000 * 8


def test_synthesize_reproduces_the_reference_generator(golden_dir, tmp_path):
    """bin/synthesize.py of this repository writes, for the same command line, the
    same file name and the same bytes as the reference's bin/synthesize.py
    (:60-294) did: fixtures tests/golden/synthesize/*.json were written by the
    reference generator itself (tests/golden/make_synthesize_fixtures.py) --
    cross / box / diffusion / hotspot, 1-D / 2-D / 3-D, forks, fractional extra
    fields, vectorize, and the 1000-stage C3 program (by digest)."""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(golden_dir.rstrip("/"))
    root = os.path.dirname(root)
    with open(os.path.join(golden_dir, "synthesize", "index.json")) as f:
        index = json.load(f)
    assert len(index) >= 15
    for case in index:
        r = subprocess.run([sys.executable, os.path.join(root, "bin", "synthesize.py")] + case["args"],
                           cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip() == "Wrote synthetic stencil to: " + case["file"], case["args"]
        with open(str(tmp_path / case["file"]), "rb") as f:
            mine = f.read()
        if "sha256" in case:
            assert len(mine) == case["bytes"] and hashlib.sha256(mine).hexdigest() == case["sha256"], case["file"]
        else:
            with open(os.path.join(golden_dir, "synthesize", case["file"]), "rb") as f:
                assert mine == f.read(), case["file"]


def test_synthesize_conventions():
    """Spot checks of the generator's conventions as Python calls (the byte-level
    comparison with the reference generator is the test above)."""
    from stencilflow_amd.programs import synthesize
    prog, name = synthesize("float32", 3, 0, 16, 16, 32, 1, 1, 1)
    assert name == "float32_3_0_16_16_32_1_1_1_0p0_2_2_cross_1.json"
    assert prog["dimensions"] == [16, 16, 32] and prog["outputs"] == ["b2"]
    assert prog["program"]["b1"]["computation_string"] == (
        "b1 = 0.16666666666666666*(b0[i-1, j, k] + b0[i+1, j, k] + "
        "b0[i, j-1, k] + b0[i, j+1, k] + b0[i, j, k-1] + b0[i, j, k+1])")
    assert prog["program"]["b0"]["boundary_conditions"] == {
        "a": {"type": "constant", "value": 0}}
    prog, _ = synthesize("float64", 4, 0.5, 64, 64, 0, 2, 1, 0,
                         fork_frequency=0.5, stencil_shape="box")
    assert set(prog["program"]) == {"b0", "b1", "b1a0", "b1a1", "b1b0", "b1b1",
                                    "b2", "b3", "b3a0"} - {"b3a0"}
    assert "a1" in prog["inputs"] and prog["program"]["b2"][
        "boundary_conditions"].keys() >= {"b1a1", "b1b1"}
    prog, _ = synthesize("float32", 2, 0, 8, 8, 8, 1, 1, 1,
                         stencil_shape="diffusion")
    assert prog["inputs"]["c6"]["input_dims"] == []
    assert prog["program"]["b0"]["computation_string"].startswith(
        "b0 = c0*a[i, j, k] + c1*a[i-1, j, k]")


def test_degenerate_programs_plan_or_fail_cleanly(tmp_path):
    """Operators without field accesses, 1x1xN and 1-D domains are planned;
    malformed descriptions raise the documented exception, not a crash."""
    import json
    import pytest
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    from tests.degenerate_programs import INVALID, VALID
    for name, prog in VALID.items():
        path = tmp_path / (name + ".json")
        path.write_text(json.dumps(prog))
        with Plan(lower(sf.KernelChainGraph(str(path)))) as plan:
            assert plan.num_launches == 1 and plan.output_names == ["b"], name
    for name, (exc, text, prog) in INVALID.items():
        path = tmp_path / (name + ".json")
        path.write_text(json.dumps(prog))
        with pytest.raises(exc, match=text):
            with Plan(lower(sf.KernelChainGraph(str(path)))):
                pass
