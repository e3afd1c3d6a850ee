"""The C-ABI library on a machine without a GPU: it loads, exports every symbol
include/sf_hip.h declares, builds plans (hipRTC cross-compiles for gfx950), and
fails loudly -- never falls back -- when asked to compute without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import stencilflow_amd as sf
from stencilflow_amd import backend, programs
from stencilflow_amd.lowering import lower

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "sf_hip.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(sf_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(backend.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == {name for name, _, _ in backend.API}
    assert backend.load_library().sf_version() == 1002


def test_plans_compile_for_all_reference_programs(programs_dir):
    for name in sorted(f[:-5] for f in os.listdir(programs_dir)
                       if f.endswith(".json")):
        chain = sf.KernelChainGraph(os.path.join(programs_dir, name + ".json"))
        with backend.Plan(lower(chain)) as plan:
            assert plan.num_launches >= 1
            assert plan.output_names == list(chain.outputs)
            assert all(n.startswith("sf_") for n in plan.kernel_names())


def test_fusion_and_buffer_reuse(tmp_path):
    path = programs.write_program(programs.jacobi3d((64, 64, 64), 10),
                                  str(tmp_path / "j.json"))
    chain = sf.KernelChainGraph(path)
    with backend.Plan(lower(chain), options={"fuse": 2}) as plan:
        assert plan.num_launches == 5 and len(plan.kernel_names()) == 1
        # input + output + two ping-pong temporaries (the reference keeps one
        # transient per stage, sdfg_generator.py:626-630)
        assert "4 device buffers" in plan.describe()
        # the innermost-dimension halo travels through the DPP data path
        src = plan.kernel_source(0)
        assert "__builtin_amdgcn_update_dpp" in src and "__shfl_up(" not in src
    with backend.Plan(lower(chain), options={"fuse": 3}) as plan:
        assert plan.num_launches == 4 and len(plan.kernel_names()) == 2
    c5 = programs.write_program(
        programs.diffusion_advection_laplacian((32, 32, 64)),
        str(tmp_path / "c5.json"))
    with backend.Plan(lower(sf.KernelChainGraph(c5)),
                      options={"fuse": 3}) as plan:
        assert plan.num_launches == 1
        assert plan.scalar_names[:2] == ["c0", "c1"]


def test_error_statuses_map_to_exceptions():
    with pytest.raises(ValueError, match="SFIR"):
        backend.Plan("not sfir")
    with pytest.raises(ValueError, match="unknown field"):
        backend.Plan("sfir 1\nprogram p\ndims 1 8\nfield b f32 1 output\n"
                     "kernel b f32\nacc x_0 x f32 none - 0\nlet float b = x_0\n"
                     "ret b\nend\n")
    with pytest.raises(RuntimeError, match="hipRTC"):
        backend.Plan("sfir 1\nprogram p\ndims 1 8\nfield a f32 1 input\n"
                     "field b f32 1 output\nkernel b f32\n"
                     "acc a_0 a f32 none - 0\nlet float b = (a_0 +* 1)\n"
                     "ret b\nend\n")


def test_no_cpu_fallback_without_a_device(programs_dir):
    lib = backend.load_library()
    if lib.sf_device_count() > 0:
        pytest.skip("a GPU is present")
    chain = sf.KernelChainGraph(
        os.path.join(programs_dir, "jacobi2d_128x128.json"))
    prog = backend.CompiledProgram(chain)
    a = np.ones((128, 128), np.float32)
    b = np.zeros((128, 128), np.float32)
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        prog(a_host=a, b_host=b)
    assert not b.any()


def _describe(tmp_path, prog, options=None):
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    with backend.Plan(lower(sf.KernelChainGraph(path)), options=options) as plan:
        return plan.describe()


def test_planner_shapes_for_the_benchmark_configurations(tmp_path):
    """The tile / chunk choices the measured numbers rest on:
    C3 fills the 256 CUs with exactly one 512-thread block each; C2 is chunked
    for about three waves per SIMD (the 2-D kernel is issue-bound below that);
    C5 fuses the three operators into one launch."""
    c3 = _describe(tmp_path, programs.jacobi3d((512, 512, 512), 4))
    m = re.search(r"star T=2 block (\d+)x(\d+) rows/thread (\d+) tiles (\d+)x(\d+) chunk (\d+)", c3)
    assert m, c3
    bx, by, rj, njt, nkt, li = map(int, m.groups())
    assert bx * 4 == 512 and nkt == 1              # whole rows, no column halo
    assert njt * nkt * -(-512 // li) == 256          # one block per CU
    assert "spill 0 scratch 0" in c3 and "agpr 0" in c3

    c2 = _describe(tmp_path, programs.jacobi2d((4096, 4096), 8))
    m = re.search(r"star T=4 block 64x1 rows/thread 1 tiles 1x(\d+) chunk (\d+)", c2)
    assert m, c2
    strips, li = map(int, m.groups())
    waves_per_simd = strips * -(-4096 // li) / 1024.0
    assert 2.5 <= waves_per_simd <= 4.0, c2

    c5 = _describe(tmp_path, programs.diffusion_advection_laplacian((512, 512, 512)))
    assert "1 launches" in c5 and "star T=3" in c5


def test_memory_instruction_modes_compile_clean(tmp_path):
    """Branch-free buffer loads/stores and the four-slot input ring (2-D and float32 3-D; float64 3-D loads straight
    into the freed window slot): the generated kernels compile for gfx950 without spills."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    import stencilflow_amd as sf

    def plan_of(prog, options=None):
        path = programs.write_program(prog, str(tmp_path / "p.json"))
        return Plan(lower(sf.KernelChainGraph(path)), options=options)

    with plan_of(programs.jacobi3d((64, 64, 128), 4)) as plan:
        text, src = plan.describe(), plan.kernel_source(0)
        assert "star T=2" in text and "spill 0 scratch 0" in text, text
        assert "#define SF_RING4 1" in src and "raw_buffer_load" in src and "raw_buffer_store" in src
    with plan_of(programs.jacobi2d((256, 512), 4)) as plan:
        assert "#define SF_RING4 1" in plan.kernel_source(0)
    with plan_of(programs.diffusion_advection_laplacian((32, 64, 128))) as plan:
        assert "#define SF_RING4 0" in plan.kernel_source(0)


def test_environment_options_and_the_frozen_option_table(tmp_path, monkeypatch):
    """SF_HIP_OPTIONS supplies defaults that the caller's options override key by key.  Round 5 froze the option table
    (sf_describe_options: 24 keys): a key that is not in it -- the measurement switches of rounds 1-4, among them
    `debug.whatif`, `experiment` and `stamp`, which built kernels with deliberately wrong results -- is refused,
    from either source, and so is a value that is not an integer."""
    from stencilflow_amd.backend import Plan, describe_options
    from stencilflow_amd.lowering import lower
    import stencilflow_amd as sf
    path = programs.write_program(programs.jacobi3d((16, 32, 64), 2), str(tmp_path / "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
    monkeypatch.setenv("SF_HIP_OPTIONS", "generic_only=1")
    with Plan(sfir) as plan:
        assert "[point]" in plan.describe()
    with Plan(sfir, options={"generic_only": 0}) as plan:
        assert "[star" in plan.describe()
    monkeypatch.setenv("SF_HIP_OPTIONS", "debug.whatif=24")
    with pytest.raises(ValueError, match="unknown plan option 'debug.whatif'"):
        Plan(sfir)
    monkeypatch.delenv("SF_HIP_OPTIONS")
    with Plan(sfir, options={"generic_only": 1}) as plan:
        assert "[point]" in plan.describe()
    keys = [line.split("=")[0] for line in describe_options().strip().splitlines()]
    assert len(keys) == len(set(keys)) <= 25 and {"fuse", "slab", "k1.bx", "dense.t2", "generic_only"} <= set(keys), keys
    for gone in ("debug.whatif", "experiment", "stamp", "k1.xlane", "k1.xbatch", "k1.pf2", "k1.skip", "k1.prio", "k1.bio",
                 "dense.il", "dense.early", "dense.stream", "generic.fast", "fuze"):
        assert gone not in keys
        with pytest.raises(ValueError, match="unknown plan option"):
            Plan(sfir, options={gone: 1})
    with pytest.raises(ValueError, match="takes an integer"):
        Plan(sfir, options="fuse=two")


def test_autotune_compiles_alternative_tile_shapes(tmp_path):
    """autotune=<k>: the first k clean tile shapes of a fused group are compiled at
    plan creation (they are timed on the device before the first execution)."""
    from stencilflow_amd.backend import Plan
    from stencilflow_amd.lowering import lower
    import stencilflow_amd as sf
    path = programs.write_program(programs.jacobi3d((96, 200, 256), 4), str(tmp_path / "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
    with Plan(sfir) as plan:
        base = len(plan.kernel_names())
    with Plan(sfir, options={"autotune": 3}) as plan:
        names = plan.kernel_names()
        assert len(names) == base + 2 and all(n.startswith("sf_star3d_f32_t2") for n in names), names
    with pytest.raises(ValueError):
        Plan(sfir, options={"autotune": 99})


def _cache_files(directory):
    return sorted(f for f in os.listdir(directory) if f.endswith(".co"))


def test_code_object_cache_round_trip_and_damage(tmp_path, monkeypatch):
    """The on-disk cache of compiled kernels (the role of `-use-cached-sdfg`,
    reference run_program.py:69-73,83-88): a second plan of the same program takes
    its objects from disk; a truncated, a garbled and a foreign file are deleted
    and the kernel is compiled again -- never handed to the loader."""
    cache = tmp_path / "cache"
    monkeypatch.setenv("SF_HIP_CACHE_DIR", str(cache))
    path = programs.write_program(programs.jacobi3d((24, 40, 72), 4, bc_value=0.375),
                                  str(tmp_path / "cached.json"))
    sfir = lower(sf.KernelChainGraph(path))

    def plan_once():
        backend.code_cache_stats(drop_process_level=True)
        before = backend.code_cache_stats()
        with backend.Plan(sfir, options={"fuse": 2}) as plan:
            resources = plan.kernel_resources()
        after = backend.code_cache_stats()
        return after[0] - before[0], after[1] - before[1], resources

    hits, compiled, first = plan_once()
    assert hits == 0 and compiled >= 1
    files = _cache_files(str(cache))
    assert len(files) == compiled
    hits, compiled2, second = plan_once()
    assert hits == compiled and compiled2 == 0 and second == first  # same objects, read from disk
    target = os.path.join(str(cache), files[0])
    good = open(target, "rb").read()
    for damaged in (good[:len(good) // 2],                       # truncated
                    good[:64] + bytes(len(good) - 64),           # payload zeroed: digest mismatch
                    b"not a code object at all"):                # foreign file
        with open(target, "wb") as f:
            f.write(damaged)
        hits, compiled3, third = plan_once()
        assert compiled3 == 1 and hits == len(files) - 1 and third == first
        assert open(target, "rb").read() == good  # rewritten by the recompile


def test_code_object_cache_can_be_switched_off(tmp_path, monkeypatch):
    monkeypatch.setenv("SF_HIP_CACHE_DIR", "off")
    path = programs.write_program(programs.jacobi2d((40, 72), 3, bc_value=0.625), str(tmp_path / "nc.json"))
    backend.code_cache_stats(drop_process_level=True)
    before = backend.code_cache_stats()
    with backend.Plan(lower(sf.KernelChainGraph(path))):
        pass
    with backend.Plan(lower(sf.KernelChainGraph(path))):
        pass
    after = backend.code_cache_stats()
    assert after[0] == before[0]  # nothing comes from disk


def test_slab_halo_must_cover_the_reach_of_a_rank_with_neighbours(tmp_path):
    """A slab that has a neighbour reads that neighbour's planes: a plan whose halo
    is shallower than a launch's reach is refused (also halo = 0), while a slab
    touching both ends of the domain needs none."""
    path = programs.write_program(programs.jacobi3d((32, 16, 32), 4), str(tmp_path / "s.json"))
    sfir = lower(sf.KernelChainGraph(path))
    for slab in ("8:16:0", "8:16:1", "0:16:1", "16:32:0"):
        with pytest.raises(ValueError, match="halo is shallower"):
            backend.Plan(sfir, options={"fuse": 2, "slab": slab})
    with backend.Plan(sfir, options={"fuse": 2, "slab": "8:16:2"}) as plan:
        assert plan.num_launches == 2
    with backend.Plan(sfir, options={"fuse": 2, "slab": "0:32:0"}) as plan:
        assert plan.num_launches == 2


def test_round3_entry_points_without_a_device(tmp_path):
    """The entry points added in round 3, as far as they go without a GPU: the RCCL id comes from
    librccl through dlopen (or the call says why it cannot), argument checks return statuses, a fresh
    plan's fused kernels carry no self-check verdict, profiling can be toggled before the first use."""
    import ctypes
    lib = backend.load_library()
    buf = ctypes.create_string_buffer(backend.HALO_RCCL_ID_BYTES)
    rc = lib.sf_halo_rccl_id(buf)
    assert rc == 0 or (rc == -2 and b"librccl" in lib.sf_last_error())
    if rc == 0:
        assert any(buf.raw)  # an ncclUniqueId is not all zeros
    assert lib.sf_halo_rccl_id(None) == -1
    assert lib.sf_halo_transport(None) is None
    assert lib.sf_halo_configure(None, 0, 0) == -1 and b"sf_halo_configure" in lib.sf_last_error()
    assert lib.sf_halo_use_rccl(None, buf, 0, 1) == -1
    assert lib.sf_self_checks_run() >= 0
    path = programs.write_program(programs.jacobi3d((24, 40, 72), 4, bc_value=0.375), str(tmp_path / "p.json"))
    with backend.Plan(lower(sf.KernelChainGraph(path))) as plan:
        assert set(plan.kernel_verdicts().values()) == {0}
        plan.set_profile(True)
        plan.set_profile(False)
        assert all(v == 0.0 for v in plan.kernel_planes().values())


def test_round4_entry_points_without_a_device():
    """sf_halo_fail / sf_halo_abandon (a bounded RCCL rung): argument checks return statuses."""
    lib = backend.load_library()
    assert lib.sf_halo_fail(None) == -1 and b"sf_halo_fail" in lib.sf_last_error()
    assert lib.sf_halo_abandon(None) == -1 and b"sf_halo_abandon" in lib.sf_last_error()


def test_fork_branches_are_planned_next_to_each_other(tmp_path):
    """The generator's fork / join program (reference bin/synthesize.py:228-253) lists the operators of its two
    branches interleaved (b3a0 b3b0 b3a1 b3b1: networkx' topological order); planned depth first each branch
    is a chain of its own and fuses: 20 launches become 14 (dag=0: without the DAG groups of round 4, which fuse
    further).  reorder=0 keeps the record's order; a plain chain is planned alike either way."""
    prog, _ = programs.synthesize("float32", 16, 0.0, 64, 64, 64, 1, 1, 1, fork_frequency=0.25)
    path = programs.write_program(prog, str(tmp_path / "fork.json"))
    sfir = lower(sf.KernelChainGraph(path))
    with backend.Plan(sfir, options={"dag": 0}) as plan:
        text = plan.describe()
        assert plan.num_launches == 14, text
        assert ": b3a0 b3a1 [star T=2" in text and ": b3b0 b3b1 [star T=2" in text
        assert plan.output_names == ["b15"]
    with backend.Plan(sfir, options={"reorder": 0, "dag": 0}) as plan:
        assert plan.num_launches == 20
    chain = programs.write_program(programs.jacobi3d((64, 64, 64), 9), str(tmp_path / "chain.json"))
    csfir = lower(sf.KernelChainGraph(chain))
    with backend.Plan(csfir) as a, backend.Plan(csfir, options={"reorder": 0}) as b:
        assert a.describe() == b.describe()


def test_compiler_identity_and_code_objects(tmp_path):
    """sf_compiler_id names the libamd_comgr the process compiles through (hipRTC binds it by soname: import order
    decides) and ends sf_plan_describe; sf_plan_kernel_object hands out the ELF the plan would load;
    sf_plan_step_kernel maps launches to compiled kernels."""
    path = programs.write_program(programs.jacobi3d((64, 64, 64), 5), str(tmp_path / "p.json"))
    with backend.Plan(lower(sf.KernelChainGraph(path))) as plan:
        ident = plan.compiler()
        assert "hiprtc" in ident and "comgr" in ident and "libamd_comgr" in ident
        assert plan.describe().rstrip().splitlines()[-1].strip() == "compiler: " + ident
        code, flags = plan.kernel_object(0)
        assert code[:4] == b"\x7fELF" and isinstance(flags, str)
        names = plan.kernel_names()
        assert [names[plan.step_kernel(s)] for s in range(plan.num_steps)] == \
            [ln.split()[1].rstrip(":") for ln in plan.describe().splitlines() if ln.strip().startswith("launch ")]
        assert plan.kernel_launch_times() == {}  # nothing profiled yet


def test_a_pinned_compiler_that_is_not_the_one_in_use_is_refused(tmp_path):
    """$SF_HIP_COMGR: a process that compiles through another libamd_comgr than the one named must not plan."""
    import subprocess
    import sys
    path = programs.write_program(programs.jacobi3d((32, 32, 32), 2), str(tmp_path / "p.json"))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import stencilflow_amd as sf\n"
            "from stencilflow_amd import backend\n"
            "from stencilflow_amd.lowering import lower\n"
            "backend.load_library()\n"
            "os.environ['SF_HIP_COMGR'] = %r\n"  # named only after the library (and its comgr) are loaded
            "try:\n"
            "    backend.Plan(lower(sf.KernelChainGraph(%r)))\n"
            "except RuntimeError as e:\n"
            "    print('refused:', e)\n") % (ROOT, "/opt/rocm/lib/libhiprtc.so", path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "refused:" in r.stdout and "SF_HIP_COMGR pins the device compiler" in r.stdout, r.stdout + r.stderr


def test_dense_and_compact_kernel_forms(tmp_path):
    """Plan-time view (no device): the generator's 125-point box -- one plain sum -- takes the dense kernel's streaming form
    with its planes arriving by LDS-DMA into a ring of two slots (round 5; six slots and staging registers in the general
    form, which an operator that is not a plain sum takes); the 27-point box takes the compact kernel under dense.t2=0."""
    big, _ = programs.synthesize("float32", 2, 0.0, 64, 64, 64, 2, 2, 2, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(big, str(tmp_path / "big.json"))))
    with backend.Plan(sfir) as plan:
        src = plan.kernel_source(0)
        assert "[dense" in plan.describe() and "#define SF_DENSE_STREAM 1" in src and "#define SFD_DLAST 2" in src
        assert "#define SF_IN_SLOTS 2" in src and "offen lds" in src and "#define SF_LAG" not in src.split("typedef")[0]
        stream_lds = plan.kernel_resources()[plan.kernel_names()[0]]["lds"]
        assert stream_lds % 2048 == 0  # (two slots of whole 1-KiB pieces)
    for k in big["program"].values():
        k["computation_string"] = k["computation_string"].replace(" + ", " - ", 1)  # (no longer a plain sum)
    sfir = lower(sf.KernelChainGraph(programs.write_program(big, str(tmp_path / "general.json"))))
    with backend.Plan(sfir) as plan:
        assert "[dense" in plan.describe() and "#define SF_DENSE_STREAM 1" not in plan.kernel_source(0)
        assert plan.kernel_resources()[plan.kernel_names()[0]]["lds"] > 2.5 * stream_lds  # (six slots)
    box, _ = programs.synthesize("float32", 2, 0.0, 64, 64, 64, 1, 1, 1, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(box, str(tmp_path / "box.json"))))
    with backend.Plan(sfir, options={"dense.t2": 0}) as plan:
        assert "[compact" in plan.describe()
        res = plan.kernel_resources()[plan.kernel_names()[0]]
        assert res["spills"] == 0 and res["scratch"] == 0, res


def test_a_kernel_family_that_fails_to_compile_is_reported(tmp_path, monkeypatch, capfd):
    """A tile shape the compiler rejects for lack of registers is skipped silently; an ERROR in a generated kernel is not:
    the planner still falls back (here: the 27-point box lands on the dense kernel, correct and slower) but says so on
    stderr.  The error is injected through SF_HIP_EXTRA_FLAGS: a macro that breaks an identifier only compact3d.h uses."""
    box, _ = programs.synthesize("float32", 2, 0.0, 48, 48, 64, 1, 1, 1, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(box, str(tmp_path / "box.json"))))
    monkeypatch.setenv("SF_HIP_CACHE_DIR", "off")
    monkeypatch.setenv("SF_HIP_EXTRA_FLAGS", "-Dsf_fetch_row=+")
    with backend.Plan(sfir) as plan:
        text = plan.describe()
    err = capfd.readouterr().err
    assert "[compact" not in text and ("[dense" in text or "[point" in text), text
    assert "warning: a compact kernel failed to COMPILE" in err, err


def test_a_short_star_chain_joins_the_compact_group_that_follows(tmp_path):
    """The generator's chains with a second spatial field in every other operator (num_fields_spatial 0.5): operator 0 is a
    star, operator 1 is not -- the star chain would end after one operator and every later group start one operator
    late (5 launches for 8 operators).  The compact kernel takes both, so the planner forms the longer group (4
    launches)."""
    prog, _ = programs.synthesize("float32", 8, 0.5, 64, 64, 64, 1, 1, 1)
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    with backend.Plan(sfir) as plan:
        text = plan.describe()
        assert text.count("\n  launch ") == 4 and text.count("[compact windows 3 T=2") == 4, text


def test_radius_three_boxes_and_crosses_take_the_streaming_dense_kernel(tmp_path):
    """The generator's box of extent 3 (343 points): a plain sum ordered by plane -- the dense kernel's streaming form with
    seven open output planes (round 4); before, the operator did not even compile on the generic kernel (342 nested
    parentheses; sources that deep now get -fbracket-depth).  A radius-3 CROSS lists its planes out of order (i-3 .. i+3,
    then j, then k): since round 5 its 12 in-plane terms join their output plane three steps after the plane arrived,
    the ring keeping three more planes (SF_LAG 3, codegen.hpp: stream_schedule) -- off the generic kernel."""
    box, _ = programs.synthesize("float32", 2, 0.0, 48, 48, 64, 3, 3, 3, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(box, str(tmp_path / "box.json"))))
    with backend.Plan(sfir) as plan:
        assert "[dense" in plan.describe() and "#define SF_ACCS 7" in plan.kernel_source(0), plan.describe()
    with backend.Plan(sfir, options={"dense": 0}) as plan:
        assert "[point]" in plan.describe()
        assert "-fbracket-depth" in plan.kernel_object(0)[1]
    cross, _ = programs.synthesize("float32", 2, 0.0, 48, 48, 64, 3, 3, 3)
    sfir = lower(sf.KernelChainGraph(programs.write_program(cross, str(tmp_path / "cross.json"))))
    with backend.Plan(sfir) as plan:
        src = plan.kernel_source(0)
        assert "[dense" in plan.describe() and "#define SF_LAG 3" in src and "#define SF_IN_SLOTS 5" in src, plan.describe()


def test_float64_boxes_take_the_fused_dense_form_or_compact_groups_two_deep(tmp_path):
    """The 27-point box in float64 at 512^3 (profiles/r04_box_f64.log): the dense kernel's fused form (256x3 threads x 3
    rows) -- 4.1e5 Mcells/s against 3.2e5 on the compact kernel two deep and 1.6e5 three deep, which the float64 default
    of fuse = 3 (the star kernel's) used to pick."""
    box, _ = programs.synthesize("float64", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(box, str(tmp_path / "box.json"))))
    with backend.Plan(sfir) as plan:
        assert "sf_dense3d_f64_t2_" in plan.describe() and "block 256x3 rows/thread 3" in plan.describe(), plan.describe()
    with backend.Plan(sfir, options={"dense.t2": 0}) as plan:
        assert "[compact windows 2 T=2" in plan.describe(), plan.describe()


def test_a_compact_group_that_mostly_recomputes_is_planned_one_operator_at_a_time(tmp_path):
    """27-point boxes with a second spatial field in every other operator: two deep the only clean tiles keep 5 of 9 or 8 of
    12 rows (and two thirds of the columns) -- 2.2-2.8e5 Mcells/s against 4.3e5 with one operator per launch
    (profiles/r04_box_extra.log).  The planner drops a group whose tile keeps less than 0.6 of what it computes and less than 0.65 of what one operator's tile keeps; an
    explicit fuse= is followed as given."""
    prog, _ = programs.synthesize("float32", 4, 0.5, 512, 512, 512, 1, 1, 1, stencil_shape="box")
    sfir = lower(sf.KernelChainGraph(programs.write_program(prog, str(tmp_path / "p.json"))))
    with backend.Plan(sfir) as plan:
        text = plan.describe()
        assert text.count("\n  launch ") == 4 and "T=2" not in text, text
    with backend.Plan(sfir, options={"fuse": 2}) as plan:
        assert "[compact windows 3 T=2" in plan.describe(), plan.describe()


def test_radius_two_crosses_pair_up_in_the_fused_dense_form(tmp_path):
    """The generator's radius-2 cross, float32, 512^3 (bench.py's `wide` workload): two operators per launch in the dense
    kernel's fused streaming form since round 5 -- each operator reaches two planes (SF_RS 2), its eight in-plane terms
    join their output plane two steps after their own plane arrived (SF_LAG / SF_LAG2 2: four input slots, four slots of
    TJ rows between the operators), rows of 34 threads so that a row of 512 is four tiles of 128 kept columns (a block
    of 1020 threads, one row each: its last wave has four lanes off and requests no pieces).  float64, 2-D and dense.t2=0 keep the wide-star kernel
    (profiles/r05_cross2_fused.log: 249 against 303 us per launch in float32, 524 against 424 in float64)."""
    cross, _ = programs.synthesize("float32", 4, 0.0, 512, 512, 512, 2, 2, 2)
    sfir = lower(sf.KernelChainGraph(programs.write_program(cross, str(tmp_path / "cross.json"))))
    with backend.Plan(sfir) as plan:
        text, src = plan.describe(), plan.kernel_source(0)
        assert "2 launches" in text and "sf_dense3d_f32_t2_" in text and "block 34x30 rows/thread 1 tiles 20x4" in text, text
        for macro in ("#define SF_RS 2\n", "#define SF_MID_HALO 0\n", "#define SF_LAG 2\n", "#define SF_LAG2 2\n", "#define SF_IN_SLOTS 4\n",
                      "#define SF_MID_SLOTS 4\n", "#define SF_ACCS 5\n", "#define SFD_DLAST 2\n", "#define SF_RCL 0\n"):
            assert macro in src, macro
        res = plan.kernel_resources()[plan.kernel_names()[0]]
        assert res["spills"] == 0 and res["scratch"] == 0 and res["vgprs"] <= 128 and res["lds"] <= 160 * 1024, res
    with backend.Plan(sfir, options={"dense.t2": 0}) as plan:
        assert "[wide star T=2" in plan.describe()
    # (sums with a factor per term too: the generator's `diffusion` shape at extent 2, 182.8 -> 144.7 us per operator)
    weighted, _ = programs.synthesize("float32", 4, 0.0, 512, 512, 512, 2, 2, 2, stencil_shape="diffusion")
    with backend.Plan(lower(sf.KernelChainGraph(programs.write_program(weighted, str(tmp_path / "diffusion.json"))))) as plan:
        assert "2 launches" in plan.describe() and "sf_dense3d_f32_t2_" in plan.describe(), plan.describe()
        assert " * (float)g" in plan.kernel_source(0)
    for dtype, dims in (("float64", (512, 512, 512)), ("float32", (4096, 4096, 0))):
        other, _ = programs.synthesize(dtype, 4, 0.0, *dims, 2, 2, 2 if dims[2] else 0)
        sfir = lower(sf.KernelChainGraph(programs.write_program(other, str(tmp_path / "other.json"))))
        with backend.Plan(sfir) as plan:
            assert "[wide star" in plan.describe() and "_t2_" not in plan.describe().replace("wstar3d_f64_t2", "").replace("wstar2d_f32_t2", ""), plan.describe()


def test_the_benchmark_chain_runs_three_operators_per_streaming_launch(tmp_path):
    """Round 5: chains of radius-1 plain sums in any order of their terms -- the benchmark's jacobi3d, a star -- run THREE
    per launch of the dense kernel's fused streaming form where a tile shape fits the grid: a second ring between the second
    and the third operator (SF_NST 3), every ring keeping the plane the in-plane terms read late (the text lists i-1, i+1
    first), rows of 34 threads, one row per thread.  jacobi3d 512^3: 91.0 against 97.9 us per operator on one box (1.47
    against 1.37e6 Mcells/s), ahead on every grid tried (profiles/r05_c3_streaming.log).  A launch of three costs about 1.5
    launches of two and a lone operator as much as two: a chain of four is two pairs on the star kernel, seven are 3 + 2 + 2;
    plans whose depth or tile the caller chose (fuse=, k1.bx ...) and dense.t2=0 keep the star kernel; small grids too."""
    def plan_of(stages, options=None, dims=(512, 512, 512)):
        sfir = lower(sf.KernelChainGraph(programs.write_program(programs.jacobi3d(dims, stages), str(tmp_path / "c3.json"))))
        return backend.Plan(sfir, options=options)
    with plan_of(7) as plan:
        text, src = plan.describe(), plan.kernel_source(0)
        assert "3 launches" in text and text.count("sf_dense3d_f32_t3_") == 1 and "[dense T=3 block 34x30 rows/thread 1" in text, text
        assert text.count("[star T=2") == 2
        for macro in ("#define SF_NST 3\n", "#define SF_MID_HALO 0\n", "#define SF_LAG 1\n", "#define SF_LAG2 1\n", "#define SF_LAG3 1\n",
                      "#define SF_IN_SLOTS 3\n", "#define SF_MID_SLOTS 3\n", "#define SF_ACCS 3\n", "#define SF_R 3\n", "#define SF_RCL 0\n"):
            assert macro in src, macro
        assert "struct sf_dense3 {" in src
        res = plan.kernel_resources()[plan.kernel_names()[0]]
        assert res["spills"] == 0 and res["scratch"] == 0 and res["vgprs"] <= 128 and res["lds"] <= 160 * 1024, res
    with plan_of(1000) as plan:  # (the benchmark: 332 launches of three, the last four operators two by two)
        names = [plan.kernel_names()[plan.step_kernel(st)] for st in range(plan.num_steps)]
        assert plan.num_launches == 334 and sum(n.startswith("sf_dense3d_f32_t3_") for n in names) == 332, plan.describe()[:300]
        assert [n.split("_")[1] for n in names[-2:]] == ["star3d", "star3d"]
    with plan_of(4) as plan:
        assert plan.describe().count("[star T=2") == 2 and "[dense" not in plan.describe()
    for options in ({"dense.t2": 0}, {"fuse": 2}, {"fuse": 3}, {"k1.bx": 128, "k1.by": 4, "k1.rj": 5}):
        with plan_of(6, options) as plan:
            assert "[star T=" in plan.describe() and "[dense" not in plan.describe(), (options, plan.describe())
    with plan_of(6, None, (24, 40, 64)) as plan:  # (a grid no tile shape fits)
        assert "[star T=2" in plan.describe() and "[dense" not in plan.describe()
    with plan_of(6, {"dense.t2": 3, "fuse": 2}) as plan:
        assert "sf_dense3d_f32_t2_" in plan.describe() and "_t3_" not in plan.describe(), plan.describe()
