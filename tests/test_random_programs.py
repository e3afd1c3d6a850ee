"""Randomly generated programs (tests/random_programs.py): the two oracle
implementations agree bit for bit (CPU), the front end lowers every program and
the library compiles it (CPU), and the HIP backend agrees with the oracle bit
for bit (GPU)."""
import json
import re

import numpy as np
import pytest

import stencilflow_amd as sf
from oracle import c_oracle, numpy_oracle as npo
from stencilflow_amd import programs
from stencilflow_amd.backend import CompiledProgram, Plan
from stencilflow_amd.lowering import lower
from tests.random_programs import random_inputs, random_program

CPU_SEEDS = list(range(100, 124))
GPU_SEEDS = list(range(100, 124))  # (tools/*_fuzz.py run hundreds more per round: profiles/r0*_fuzz*.log)


@pytest.fixture(autouse=True)
def _no_plan_time_self_check(monkeypatch):
    """Hundreds of one-off programs: the plan-time self-check (one hipRTC compilation per distinct
    operator for its reference kernels) would double this module's time, and what it checks -- fused
    against generic kernels -- these tests check against the oracle anyway."""
    monkeypatch.setenv("SF_HIP_SELF_CHECK", "0")


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


STAR_CPU_SEEDS = list(range(0, 12))
STAR_GPU_SEEDS = list(range(0, 18))


def _star_case(seed, tmp_path):
    from tests.random_programs import star_program
    prog = star_program(seed)
    rng = np.random.default_rng(seed + 7)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path), {"fuse": int(rng.integers(1, 5))}


@pytest.mark.parametrize("seed", STAR_CPU_SEEDS)
def test_random_star_chains_plan(seed, tmp_path):
    """Every random star chain is planned onto fused star kernels (CPU: hipRTC
    only) and the two oracles agree on it."""
    prog, ins, chain, opt = _star_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options=opt) as plan:
        assert "[star" in plan.describe()
        assert sorted(plan.output_names) == sorted(prog["outputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", STAR_GPU_SEEDS)
def test_hip_matches_oracle_on_random_star_chains(seed, tmp_path):
    prog, ins, chain, opt = _star_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options=opt) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:400])


COPY_SEEDS = list(range(0, 8))  # (tools/star_fuzz.py --copy: profiles/r03_copy_fuzz*.log)


def _copy_case(seed, tmp_path, generator="star_program"):
    """A random star chain with `copy` boundaries (tests/random_programs.py: with_copy_boundaries).
    The reference's CPU expansion -- and with it the oracles -- has no `copy` (stencil/cpu.py:87
    raises), so the fused star kernel is checked against the product's own generic kernel, one
    operator per launch (the form tests/test_gpu_parity.py::test_copy_boundary_condition pins
    against a NumPy statement of the rule)."""
    import tests.random_programs as rp
    prog = rp.with_copy_boundaries(getattr(rp, generator)(seed), seed)
    rng = np.random.default_rng(seed + 7)
    ins = {}
    for name, desc in prog["inputs"].items():
        dims = desc.get("input_dims", ["i", "j", "k"][3 - len(prog["dimensions"]):])
        shape = [prog["dimensions"][["i", "j", "k"][3 - len(prog["dimensions"]):].index(d)] for d in dims]
        ins[name] = (rng.uniform(-1, 1, shape).astype(npo._NP[desc["data_type"]]) if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path), {"fuse": int(rng.integers(1, 5))}


def _run_plan(chain, prog, ins, options):
    with Plan(lower(chain), options=options) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        return dict(zip(plan.output_names, outs)), plan.describe()


@pytest.mark.parametrize("seed", COPY_SEEDS[:6])
def test_star_chains_with_copy_boundaries_are_fused(seed, tmp_path):
    prog, ins, chain, opt = _copy_case(seed, tmp_path)
    with Plan(lower(chain), options=opt) as plan:
        assert "[star" in plan.describe() and "[point" not in plan.describe()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", COPY_SEEDS)
def test_fused_copy_boundaries_match_the_generic_kernel(seed, tmp_path):
    prog, ins, chain, opt = _copy_case(seed, tmp_path)
    want, desc = _run_plan(chain, prog, ins, {"generic_only": 1})
    assert "[star" not in desc
    got, desc = _run_plan(chain, prog, ins, opt)
    assert "[star" in desc
    for n in want:
        assert np.array_equal(got[n], want[n], equal_nan=True), (seed, n, desc[:400])


@pytest.mark.gpu
@pytest.mark.parametrize("generator,kernel,seed", [("wide_program", "[wide star", 0), ("wide_program", "[wide star", 1),
                                                   ("compact_program", "[compact", 0), ("dense_program", "[dense", 3)])
def test_copy_boundaries_in_the_other_fused_kernels_match_the_generic_kernel(generator, kernel, seed, tmp_path):
    prog, ins, chain, opt = _copy_case(seed, tmp_path, generator)
    opt = {"fuse": min(opt["fuse"], 3)}
    want, desc = _run_plan(chain, prog, ins, {"generic_only": 1})
    got, desc = _run_plan(chain, prog, ins, opt)
    assert kernel in desc or "[star" in desc
    for n in want:
        assert np.array_equal(got[n], want[n], equal_nan=True), (generator, seed, n, desc[:400])


@pytest.mark.gpu
@pytest.mark.parametrize("generator,seed", [("star_program", 600), ("star_program", 601), ("wide_program", 602),
                                            ("compact_program", 603), ("dense_program", 605)])
def test_fused_copy_boundaries_under_slab_decomposition(generator, seed, tmp_path):
    """`copy` is decided at GLOBAL coordinates: a slab's first plane is a boundary only on rank 0."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    prog, ins, chain, opt = _copy_case(seed, tmp_path, generator)
    opt = {"fuse": min(opt["fuse"], 3)}
    if len(prog["dimensions"]) == 3:
        prog["dimensions"][0] = max(prog["dimensions"][0], 48)
    else:
        prog["dimensions"][0] = max(prog["dimensions"][0], 96)
    rng = np.random.default_rng(seed + 13)
    for name, desc in prog["inputs"].items():
        if not np.isscalar(ins[name]) and ins[name].shape[0] != prog["dimensions"][0] and ins[name].ndim == len(prog["dimensions"]):
            ins[name] = rng.uniform(-1, 1, prog["dimensions"]).astype(ins[name].dtype)
    chain = sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "q.json")))
    want, _ = _run_plan(chain, prog, ins, {"generic_only": 1})
    sfir = lower(chain)
    shape, world = tuple(prog["dimensions"]), int(rng.integers(2, 4))
    exch = LocalExchanger(world)
    groups = int(rng.integers(1, 3))
    runners = [SlabRunner(sfir, shape, r, world, options=opt, exchanger=exch.for_rank(r), groups_per_exchange=groups)
               for r in range(world)]
    assert "[point" not in runners[0].plan.describe() or generator == "dense_program"
    for r in runners:
        if r.plan.scalar_names:
            r.plan.set_scalars([ins[n] for n in r.plan.scalar_names])
        r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi] if ins[n].ndim == len(shape) else ins[n])
                  for n in r.plan.input_names])
    run_lockstep(runners)
    for oi, name in enumerate(runners[0].plan.output_names):
        got = np.zeros(shape, dtype=want[name].dtype)
        for r in runners:
            parts = [np.zeros(r.local_shape, dtype=want[n].dtype) for n in r.plan.output_names]
            r.download(parts)
            got[r.lo:r.hi] = parts[oi]
        assert np.array_equal(got, want[name], equal_nan=True), (seed, name)
    for r in runners:
        r.close()


WIDE_CPU_SEEDS = list(range(0, 8))
WIDE_GPU_SEEDS = list(range(0, 16))


def _wide_case(seed, tmp_path):
    from tests.random_programs import wide_program
    prog = wide_program(seed)
    rng = np.random.default_rng(seed + 11)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path), {"fuse": int(rng.integers(1, 4))}


@pytest.mark.parametrize("seed", WIDE_CPU_SEEDS)
def test_random_wide_star_chains_plan(seed, tmp_path):
    """Random chains of radius-2 stars (kernels/wstar3d.h; what the reference's generator emits
    for an extent of 2, bin/synthesize.py:19-31) are planned onto fused wide-star launches (CPU:
    hipRTC only) and the two oracles agree on them."""
    prog, ins, chain, opt = _wide_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options=opt) as plan:
        import re
        far = any(re.search(r"[ijk][+-]2\b", k["computation_string"]) for k in prog["program"].values())
        assert ("[wide star" in plan.describe()) == far, plan.describe()
        assert sorted(plan.output_names) == sorted(prog["outputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", WIDE_GPU_SEEDS)
def test_hip_matches_oracle_on_random_wide_star_chains(seed, tmp_path):
    prog, ins, chain, opt = _wide_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options=opt) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:400])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(300, 308)))
def test_random_wide_star_chains_under_slab_decomposition(seed, tmp_path):
    """The same chains on two or three in-process slabs of unequal height (a launch of T fused
    radius-2 operators reaches 2 T planes across a slab boundary)."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from tests.random_programs import wide_program
    prog = wide_program(seed)
    prog["dimensions"][0] = max(prog["dimensions"][0], 40)
    rng = np.random.default_rng(seed + 13)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    want = npo.run_reference(prog, inputs=ins)
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    shape, world = tuple(prog["dimensions"]), int(rng.integers(2, 4))
    exch = LocalExchanger(world)
    fuse, groups = int(rng.integers(1, 3)), int(rng.integers(1, 3))  # (alike on all ranks)
    runners = [SlabRunner(sfir, shape, r, world, options={"fuse": fuse}, exchanger=exch.for_rank(r),
                          groups_per_exchange=groups) for r in range(world)]
    for r in runners:
        if r.plan.scalar_names:
            r.plan.set_scalars([ins[n] for n in r.plan.scalar_names])
        r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi]) for n in r.plan.input_names])
    run_lockstep(runners)
    for oi, name in enumerate(runners[0].plan.output_names):
        got = np.zeros(shape, dtype=want[name].dtype)
        for r in runners:
            parts = [np.zeros(r.local_shape, dtype=want[n].dtype) for n in r.plan.output_names]
            r.download(parts)
            got[r.lo:r.hi] = parts[oi]
        assert np.array_equal(got, want[name], equal_nan=True), (seed, name)
    for r in runners:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,cols,options", [(403, 520, {"fuse": 2}), (408, 776, {"fuse": 2}), (419, 520, {"fuse": 3}),
                                                (428, 520, {"fuse": 1, "k1.bx": 128, "k1.by": 2, "k1.rj": 4}),
                                                (418, 1032, {"fuse": 1, "k1.bx": 256, "k1.by": 2, "k1.rj": 3}),
                                                (425, 520, {"fuse": 2, "k1.bx": 128, "k1.by": 4, "k1.rj": 3}),
                                                (423, 300, {"fuse": 2}), (409, 1100, {"fuse": 2})])
def test_wide_star_chains_on_rows_wider_than_a_tile(seed, cols, options, tmp_path):
    """Rows that are cut into several k-tiles of equal useful width (kernels/wstar3d.h: SF_TKI) and
    thread rows of two and four waves, whose edge elements travel through LDS with virtual waves at
    the row ends -- the paths the small random grids never reach."""
    from tests.random_programs import wide_program
    prog = wide_program(seed)
    if len(prog["dimensions"]) == 3:
        prog["dimensions"] = [max(9, min(prog["dimensions"][0], 14)), max(10, min(prog["dimensions"][1], 26)), cols]
    else:
        prog["dimensions"] = [max(12, min(prog["dimensions"][0], 40)), cols]
    rng = np.random.default_rng(seed + 17)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    want = npo.run_reference(prog, inputs=ins)
    chain = sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json")))
    if len(prog["dimensions"]) == 2:
        options = {k: v for k, v in options.items() if k == "fuse"}
    with Plan(lower(chain), options=options) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


DENSE_CPU_SEEDS = list(range(0, 6))
DENSE_GPU_SEEDS = list(range(0, 4))  # tools/star_fuzz.py --generator dense: profiles/r03_dense_fuzz.log


def _dense_case(seed, tmp_path, generator="dense_program"):
    import tests.random_programs as rp
    prog = getattr(rp, generator)(seed)
    rng = np.random.default_rng(seed + 19)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path)


@pytest.mark.parametrize("seed", DENSE_CPU_SEEDS)
def test_random_dense_chains_plan(seed, tmp_path):
    """Operators with dense radius-2 neighbourhoods (kernels/dense3d.h; the generator's box of
    extent 2, bin/synthesize.py:19-31) are planned onto LDS-tiled dense launches (CPU: hipRTC only)
    and the two oracles agree on them."""
    prog, ins, chain = _dense_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain)) as plan:
        # (an operator whose dense form does not come out clean -- many double-typed terms -- keeps the generic kernel)
        assert plan.describe().count("[dense") >= 1, plan.describe()
        assert plan.describe().count("[dense") + plan.describe().count("[point]") == len(prog["program"]), plan.describe()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", DENSE_GPU_SEEDS)
def test_hip_matches_oracle_on_random_dense_chains(seed, tmp_path):
    prog, ins, chain = _dense_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain)) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:400])


@pytest.mark.parametrize("seed", [0, 3, 4])
def test_plain_sums_take_the_dense_kernels_sum_form(seed, tmp_path):
    """Operators that are one left-associated sum of accesses, the terms in any order (tests/random_programs.py:
    dense_sum_program): the oracles agree, and the dense launches stream (SF_DENSE_STREAM 1 in the generated source: a
    term joins its output plane at the step its plane and every earlier term have arrived, codegen.hpp: stream_schedule)."""
    prog, ins, chain = _dense_case(seed, tmp_path, "dense_sum_program")
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain)) as plan:
        names = plan.kernel_names()
        dense = [i for i, n in enumerate(names) if n.startswith("sf_dense") and n in plan.describe()]
        assert dense and all("#define SF_DENSE_STREAM 1" in plan.kernel_source(i) for i in dense), plan.describe()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 8])
def test_hip_matches_oracle_on_random_plain_sums(seed, tmp_path):
    prog, ins, chain = _dense_case(seed, tmp_path, "dense_sum_program")
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain)) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:400])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(500, 503)))
def test_random_dense_chains_under_slab_decomposition(seed, tmp_path):
    """The same on two or three in-process slabs (a dense launch reaches up to two planes across a
    slab boundary; the runner exchanges what every launch reads)."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from tests.random_programs import dense_program
    prog = dense_program(seed)
    prog["dimensions"][0] = max(prog["dimensions"][0], 24)
    rng = np.random.default_rng(seed + 23)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    want = npo.run_reference(prog, inputs=ins)
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    shape, world = tuple(prog["dimensions"]), int(rng.integers(2, 4))
    exch = LocalExchanger(world)
    groups = int(rng.integers(1, 3))
    runners = [SlabRunner(sfir, shape, r, world, exchanger=exch.for_rank(r), groups_per_exchange=groups) for r in range(world)]
    for r in runners:
        if r.plan.scalar_names:
            r.plan.set_scalars([ins[n] for n in r.plan.scalar_names])
        r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi]) for n in r.plan.input_names])
    run_lockstep(runners)
    name = runners[0].plan.output_names[0]
    got = np.zeros(shape, dtype=want[name].dtype)
    for r in runners:
        parts = [np.zeros(r.local_shape, dtype=want[n].dtype) for n in r.plan.output_names]
        r.download(parts)
        got[r.lo:r.hi] = parts[0]
    assert np.array_equal(got, want[name], equal_nan=True), seed
    for r in runners:
        r.close()


COMPACT_CPU_SEEDS = list(range(0, 6))
COMPACT_GPU_SEEDS = list(range(0, 6))  # (tools/compact_fuzz.py, tools/star_fuzz.py --generator compact: profiles/)


def _compact_case(seed, tmp_path):
    from tests.random_programs import compact_program
    prog = compact_program(seed)
    rng = np.random.default_rng(seed + 11)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path), {"fuse": int(rng.integers(1, 5 if len(prog["dimensions"]) == 2 else 4))}


@pytest.mark.parametrize("seed", COMPACT_CPU_SEEDS)
def test_random_compact_chains_plan(seed, tmp_path):
    """Random chains of compact operators (27-point neighbourhoods, extra streamed
    fields) are planned onto plane-streaming kernels -- kernels/compact3d.h, or
    star3d.h where every operator of the chain happens to be a star -- never the
    generic kernel (CPU: hipRTC only), and the two oracles agree on them."""
    prog, ins, chain, opt = _compact_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options=opt) as plan:
        text = plan.describe()
        assert "[compact" in text or "[star" in text, text
        assert "[point]" not in text, text
        assert sorted(plan.output_names) == sorted(prog["outputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", COMPACT_GPU_SEEDS)
def test_hip_matches_oracle_on_random_compact_chains(seed, tmp_path):
    prog, ins, chain, opt = _compact_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options=opt) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


def _box_sum_case(seed, tmp_path):
    import tests.random_programs as rp
    prog = rp.box_sum_program(seed)
    rng = np.random.default_rng(seed + 7)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path)


@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_radius_one_plain_sums_pair_up_in_the_dense_kernel(seed, tmp_path):
    """Chains of plain sums over subsets of {-1,0,1}^d ordered by plane (tests/random_programs.py: box_sum_program): the
    oracles agree, and with dense.t2=2 (any grid, not only those a tile shape fits well) consecutive pairs share one
    streaming dense launch (CPU: hipRTC only); dense.t2=0 plans none."""
    prog, ins, chain = _box_sum_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options={"dense.t2": 2}) as plan:
        text = plan.describe()
        pairs = [i for i, n in enumerate(plan.kernel_names()) if n.startswith("sf_dense") and "_t2_" in n and n in text]
        assert pairs, text
        assert all("#define SF_DENSE_T2 1" in plan.kernel_source(i) for i in pairs)
        assert sorted(plan.output_names) == sorted(prog["outputs"])
    with Plan(lower(chain), options={"dense.t2": 0}) as plan:
        assert "_t2_" not in "".join(n for n in plan.kernel_names() if n.startswith("sf_dense"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 6)))
def test_hip_matches_oracle_on_fused_pairs_of_plain_sums(seed, tmp_path):
    prog, ins, chain = _box_sum_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options={"dense.t2": 2}) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 3, 4, 5, 8])
def test_hip_matches_oracle_on_plain_sums_three_per_launch(seed, tmp_path):
    """The same chains under dense.t2=3, fuse=3: up to three operators per launch, sums in random order included (they
    end a fused group otherwise)."""
    prog, ins, chain = _box_sum_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options={"dense.t2": 3, "fuse": 3}) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


def _weighted_cross_case(seed, tmp_path):
    import tests.random_programs as rp
    prog = rp.weighted_cross_program(seed)
    rng = np.random.default_rng(seed + 7)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path)


@pytest.mark.parametrize("seed", [0, 1, 5])
def test_weighted_crosses_take_the_fused_streaming_forms(seed, tmp_path):
    """Chains of crosses of radius 1 or 2 with a factor per term -- literal or scalar, before or after the access, the
    generator's `diffusion` order or shuffled (tests/random_programs.py: weighted_cross_program): the oracles agree, and
    under dense.t2=3 consecutive float32 3-D operators share a launch of the dense kernel's fused form whose functors
    multiply term by term (CPU: hipRTC only)."""
    prog, ins, chain = _weighted_cross_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options={"dense.t2": 3}) as plan:
        text = plan.describe()
        fused = [i for i, n in enumerate(plan.kernel_names()) if n.startswith("sf_dense") and ("_t2_" in n or "_t3_" in n) and n in text]
        assert fused and any(" * (" in plan.kernel_source(i) for i in fused), text


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 8)))
def test_hip_matches_oracle_on_fused_weighted_crosses(seed, tmp_path):
    prog, ins, chain = _weighted_cross_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options={"dense.t2": 3}) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


def _sparse_sum_case(seed, tmp_path):
    import tests.random_programs as rp
    prog = rp.sparse_sum_program(seed)
    rng = np.random.default_rng(seed + 7)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    return prog, ins, sf.KernelChainGraph(path)


@pytest.mark.parametrize("seed", [0, 3, 4])
def test_sparse_radius_two_sums_pair_up_in_the_dense_kernel(seed, tmp_path):
    """Chains of plain sums of at most 16 terms within two points, in the generator's cross order, shuffled, or random
    subsets in random order (tests/random_programs.py: sparse_sum_program): the oracles agree, and with dense.t2=2
    consecutive float32 3-D pairs share one streaming dense launch whose rings keep the planes the late terms read (CPU:
    hipRTC only); dense.t2=0 plans none."""
    prog, ins, chain = _sparse_sum_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options={"dense.t2": 2}) as plan:
        text = plan.describe()
        pairs = [i for i, n in enumerate(plan.kernel_names()) if n.startswith("sf_dense") and "_t2_" in n and n in text]
        assert pairs, text
        reach2 = [i for i in pairs if "#define SF_RS 2\n" in plan.kernel_source(i)]
        assert reach2, text
        for i in reach2:
            src = plan.kernel_source(i)
            slots = {m: int(re.search(r"#define %s (\d+)\n" % m, src).group(1)) if re.search(r"#define %s (\d+)\n" % m, src) else d
                     for m, d in (("SF_IN_SLOTS", 0), ("SF_LAG", 0), ("SF_MID_SLOTS", 2), ("SF_LAG2", 0))}
            assert slots["SF_IN_SLOTS"] == 2 + slots["SF_LAG"] and slots["SF_MID_SLOTS"] == 2 + slots["SF_LAG2"], slots
        assert sorted(plan.output_names) == sorted(prog["outputs"])
    with Plan(lower(chain), options={"dense.t2": 0}) as plan:
        assert "_t2_" not in "".join(n for n in plan.kernel_names() if n.startswith("sf_dense"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 8)))
def test_hip_matches_oracle_on_fused_pairs_of_sparse_radius_two_sums(seed, tmp_path):
    prog, ins, chain = _sparse_sum_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options={"dense.t2": 2}) as plan:
        if plan.scalar_names:
            plan.set_scalars([ins[n] for n in plan.scalar_names])
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]]) for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, plan.describe()[:600])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(200, 206)))
def test_random_compact_chains_under_slab_decomposition(seed, tmp_path):
    """Compact chains (incl. stages with an extra streamed field, which the runner
    exchanges like any slab-split field a launch reads across planes) split into 2-3
    unequal slabs, all ranks in this process on the one GPU: equal to the oracle bit
    for bit."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    from tests.random_programs import compact_program
    prog = compact_program(seed)
    rng = np.random.default_rng(seed + 13)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    want = npo.run_reference(prog, inputs=ins)
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
    shape = tuple(prog["dimensions"])
    fuse = int(rng.integers(1, 3))
    world = 2 if shape[0] < 4 * 3 * fuse else int(rng.integers(2, 4))
    if shape[0] < 2 * world * 2 * fuse:
        pytest.skip("outermost extent too small to split")
    exch = LocalExchanger(world)
    groups = int(rng.integers(1, 3))  # (alike on all ranks)
    runners = [SlabRunner(sfir, shape, r, world, options={"fuse": fuse}, exchanger=exch.for_rank(r),
                          groups_per_exchange=groups) for r in range(world)]
    slabbed = {n for n, d in p["inputs"].items() if npo._input_dims(p, n) and npo._input_dims(p, n)[0] == npo._own_iterators(p)[0]}
    for r in runners:
        if r.plan.scalar_names:
            r.plan.set_scalars([ins[n] for n in r.plan.scalar_names])
        r.upload([np.ascontiguousarray(ins[n][r.lo:r.hi]) if n in slabbed else ins[n] for n in r.plan.input_names])
    run_lockstep(runners)
    for out_index, name in enumerate(runners[0].plan.output_names):
        got = np.zeros(shape, dtype=want[name].dtype)
        for r in runners:
            parts = [np.zeros(r.local_shape, dtype=want[n].dtype) for n in r.plan.output_names]
            r.download(parts)
            got[r.lo:r.hi] = parts[out_index]
        assert np.array_equal(got, want[name], equal_nan=True), (seed, name, runners[0].plan.describe()[:500])
    for r in runners:
        r.close()


# ---- DAG groups of kernels/star3d.h (round 4): forks, joins, intermediates with several readers ---------------------
DAG_CPU_SEEDS = list(range(700, 706))
DAG_GPU_SEEDS = list(range(700, 716))  # (tools/star_fuzz.py --generator dag carries the volume: profiles/r04_dag_fuzz*.log)


def _dag_case(seed, tmp_path):
    from tests.random_programs import dag_program
    prog = dag_program(seed)
    rng = np.random.default_rng(seed + 7)
    p = npo.load_program(prog)
    ins = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        ins[name] = (rng.uniform(-1, 1, npo._dims_shape(p, dims)).astype(npo._NP[desc["data_type"]])
                     if dims else desc["data"])
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    # 3-D programs get room for DAG groups every second seed (by default they hold as many windows as a chain)
    opt = {"fuse": int(rng.integers(2, 5))}
    if seed % 2 == 0:
        opt["dag.windows"] = int(rng.integers(3, 6))
    return prog, ins, sf.KernelChainGraph(path), opt


@pytest.mark.parametrize("seed", DAG_CPU_SEEDS)
def test_random_dag_programs_plan(seed, tmp_path):
    """Fork / join programs are planned (CPU: hipRTC only) with every output named, and the two oracles agree."""
    prog, ins, chain, opt = _dag_case(seed, tmp_path)
    a = npo.run_reference(prog, inputs=ins)
    b = c_oracle.CompiledReference(prog).run(inputs=ins)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), (seed, k)
    with Plan(lower(chain), options=opt) as plan:
        assert sorted(plan.output_names) == sorted(prog["outputs"])
        for s in range(plan.num_steps):
            assert 1 <= len(plan.step_outputs(s)) <= 4 and plan.step_outputs(s)[0] == plan.step_output(s)


def test_dag_groups_form_where_they_save_field_passes(tmp_path):
    """2-D fork: the two branches, the join and what follows become one launch; 3-D holds one register window more
    than a chain and evaluates both branches of a fork from one read of the forked field (dag.windows=2: as many
    as a chain, no group forms).  dag=0 turns the machinery off."""
    prog2, _ = programs.synthesize("float32", 16, 0.0, 256, 512, 0, 1, 1, 0, fork_frequency=0.25)
    sfir2 = lower(sf.KernelChainGraph(programs.write_program(prog2, str(tmp_path / "f2.json"))))
    with Plan(sfir2) as plan:
        text = plan.describe()
        assert "[dag: 6 stages, 6 windows]" in text and "[dag: 6 stages, 5 windows]" in text, text
        two = [s for s in range(plan.num_steps) if len(plan.step_outputs(s)) == 2]
        assert len(two) >= 1  # the group that ends in two branch ends materialises both
        launches = plan.num_launches
    with Plan(sfir2, options={"dag": 0}) as plan:
        assert "[dag:" not in plan.describe() and plan.num_launches > launches
    prog3, _ = programs.synthesize("float32", 16, 0.0, 64, 64, 64, 1, 1, 1, fork_frequency=0.25)
    sfir3 = lower(sf.KernelChainGraph(programs.write_program(prog3, str(tmp_path / "f3.json"))))
    with Plan(sfir3, options={"dag.windows": 2}) as plan:
        assert "[dag:" not in plan.describe() and plan.num_launches == 14
    with Plan(sfir3) as plan:  # (default in 3-D: one window more than a chain of the same depth)
        assert plan.describe().count("[dag: 4 stages, 3 windows]") == 3 and plan.num_launches == 11


@pytest.mark.gpu
@pytest.mark.parametrize("seed", DAG_GPU_SEEDS)
def test_hip_matches_oracle_on_random_dag_programs(seed, tmp_path):
    prog, ins, chain, opt = _dag_case(seed, tmp_path)
    want = npo.run_reference(prog, inputs=ins)
    with Plan(lower(chain), options=opt) as plan:
        outs = [np.zeros(prog["dimensions"], dtype=npo._NP[prog["program"][n]["data_type"]])
                for n in plan.output_names]
        plan.run([np.ascontiguousarray(ins[n]) for n in plan.input_names], outs, 1)
        for n, got in zip(plan.output_names, outs):
            assert np.array_equal(got, want[n], equal_nan=True), (seed, n, opt, plan.describe()[:600])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(720, 726)))
def test_random_dag_programs_under_slab_decomposition(seed, tmp_path):
    """DAG groups on two or three in-process slabs of unequal height: a launch reaches as many planes as the group
    is deep, and a group that materialises several fields leaves all of them to be exchanged."""
    from stencilflow_amd.distributed import LocalExchanger, SlabRunner, run_lockstep
    prog, ins, chain, opt = _dag_case(seed, tmp_path)
    prog["dimensions"][0] = max(prog["dimensions"][0], 40)
    p = npo.load_program(prog)
    rng = np.random.default_rng(seed + 13)
    ins = {name: rng.uniform(-1, 1, npo._dims_shape(p, npo._input_dims(p, name))).astype(npo._NP[desc["data_type"]])
           for name, desc in p["inputs"].items()}
    want = npo.run_reference(prog, inputs=ins)
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    shape, world = tuple(prog["dimensions"]), int(rng.integers(2, 4))
    exch = LocalExchanger(world)
    runners = [SlabRunner(sfir, shape, r, world, options=opt, exchanger=exch.for_rank(r), groups_per_exchange=1)
               for r in range(world)]
    split = "i" if len(shape) == 3 else "j"
    for r in runners:
        local = []
        for name in r.plan.input_names:
            idims = npo._input_dims(p, name)
            arr = np.ascontiguousarray(ins[name])
            local.append(np.ascontiguousarray(arr[r.lo:r.hi]) if idims and idims[0] == split else arr)
        r.upload(local)
    run_lockstep(runners)
    for oi, name in enumerate(runners[0].plan.output_names):
        got = np.zeros(shape, dtype=want[name].dtype)
        for r in runners:
            parts = [np.zeros(r.local_shape, dtype=want[n].dtype) for n in r.plan.output_names]
            r.download(parts)
            got[r.lo:r.hi] = parts[oi]
        assert np.array_equal(got, want[name], equal_nan=True), (seed, name, opt)
    for r in runners:
        r.close()
