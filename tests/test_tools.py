"""CPU checks of the measurement tools whose output DESIGN.md quotes: tools/isa_stats.py (instruction budget per cell
update of the product's code objects) and tools/profile_table.py (the measured table regenerated from profiles/)."""
import importlib.util
import os

from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
import stencilflow_amd as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_instruction_budget_of_the_star_kernel(tmp_path):
    """The jacobi3d operator needs 5 adds + 1 multiply in double per update and its conversions: the loop of the
    compiled T = 2 kernel must show them (arith between 6 and 7 per update, nothing spilled to scratch)."""
    isa = _load("isa_stats")
    path = programs.write_program(programs.jacobi3d((512, 512, 512), 4), str(tmp_path / "c3.json"))
    with Plan(lower(sf.KernelChainGraph(path))) as plan:
        rec = isa.budget(plan, 0)
    assert rec["kernel"].startswith("sf_star3d_f32_t2_") and rec["updates_per_thread_and_iteration"] == 160
    pu = rec["per_update"]
    assert 6.0 <= pu["arith"] <= 7.0 and 3.5 <= pu["cvt"] <= 5.5, pu
    assert pu["all_valu"] < 18 and pu["vmem"] < 0.8, pu  # (both arms of the wave-uniform load-policy branch are counted)
    assert isa.classify("v_pk_add_f32") == "arith" and isa.classify("v_mov_b32_dpp") == "dpp+mov"
    assert isa.classify("s_nop") == "wait" and isa.classify("ds_read_b128") == "lds"
    # the loop is a LOOP: shorter than the kernel (round 4 found the finder counting the whole kernel)
    assert rec["loop_instructions"] < rec["instructions"] - 200, (rec["loop_instructions"], rec["instructions"])


def test_loop_finder_reads_backward_branches_as_llvm_objdump_prints_them():
    """llvm-objdump prints the 16-bit offset of a SOPP branch unsigned: a backward branch over 650 instructions reads
    `s_cbranch_scc0 64886`, not `-650`.  The loop is the backward branch spanning the most instructions."""
    isa = _load("isa_stats")
    insts = [(0x1000 + 4 * i, "v_add_f32_e32", "v1, v2, v1") for i in range(700)]
    insts[10] = (0x1000 + 40, "s_cbranch_execz", "5")                      # forward, short
    insts[690] = (0x1000 + 4 * 690, "s_cbranch_scc0", str(65536 - 651))    # back to instruction 40
    insts[300] = (0x1000 + 1200, "s_cbranch_vccnz", str(65536 - 21))       # a short inner loop
    assert isa.main_loop(insts) == (40, 690)
    assert isa.main_loop(insts[:600]) == (280, 300)
    assert isa.main_loop(insts[:100]) is None


def test_measured_table_is_generated_from_the_committed_profiles():
    table = _load("profile_table")
    rows = table.rows("r04") or table.rows("r03")
    assert rows, "no committed kernel statistics"
    c3 = [r for r in rows if r["workload"].startswith("C3")]
    assert c3 and c3[0]["launches"] >= 500 and 150 < c3[0]["avg_us"] < 260
    if c3[0]["pmc_bytes"]:
        assert 1.0 <= c3[0]["over_compulsory"] <= 1.1 and 0.5 < c3[0]["frac"] < 0.8
    text = table.table("r04" if table.rows("r04") else "r03")
    assert text.splitlines()[0].startswith("| workload | kernel |")
