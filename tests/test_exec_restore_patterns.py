"""The classifier behind `count_late_exec_restores` (stencilflow_amd/csrc/codecache.cpp) on
hand-made instruction patterns: the compiler's assembly of a small kernel gets a few
instructions appended behind its `s_endpgm` (never executed, but part of the code the
library reads), is assembled, and handed to the library in place of the compiler's object
(SF_HIP_OBJECT_DIR).  Compile-only: no GPU needed."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

PROGRAM = r"""
import json, os, sys, tempfile
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
with tempfile.TemporaryDirectory() as tmp:
    path = programs.write_program(programs.jacobi3d((6, 7, 8), 1), os.path.join(tmp, "p.json"))
    plan = Plan(lower(sf.KernelChainGraph(path)), options="generic_only=1")
    name = plan.kernel_names()[0]
    print("RESULT " + json.dumps({"name": name, "source": plan.kernel_source(0), "resources": plan.kernel_resources()[name],
                                  "describe": plan.describe(), "all": plan.kernel_resources()}))
"""

PATTERNS = {
    # the fault: a vector copy and an SGPR split copy between the `if` body and the EXEC restore
    "split_copy": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n v_mov_b32_e32 v3, v4\n"
                   " s_mov_b64 s[4:5], s[10:11]\n s_or_b64 exec, exec, s[0:1]\n", 1),
    # spill-lane access ahead of the restore
    "lane_reload": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n v_readlane_b32 s12, v255, 3\n"
                    " s_or_b64 exec, exec, s[0:1]\n", 1),
    # a vector spill store between the scalar copy and the restore
    "copy_then_store": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n s_mov_b64 vcc, s[6:7]\n"
                        " scratch_store_dwordx4 off, v[4:7], off offset:16\n s_or_b64 exec, exec, s[0:1]\n", 1),
    # a rematerialised constant ahead of the restore, behind other instructions of the body
    "constant_after_body": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n s_mov_b32 s6, 0x3fdc28f5\n"
                            " s_or_b64 exec, exec, s[0:1]\n", 1),
    # an `else` body that only selects a constant reads the same as a rematerialised constant next to
    # allocator copies: flagged when scalar registers are exhausted (the price of being sure), not
    # searched at all below that (test_objects_far_from_register_exhaustion_are_not_searched)
    "constant_select": ("s_andn2_saveexec_b64 s[0:1], s[6:7]\n s_mov_b32 s6, 0xc28f5c29\n s_mov_b32 s7, 0x3fdc28f5\n"
                        " v_mov_b64_e32 v[6:7], s[6:7]\n s_or_b64 exec, exec, s[0:1]\n", 1),
    # seen in a failing object (tools/config_fuzz.py, math program): allocator code inside an `else` prologue
    "else_prologue": ("s_or_saveexec_b64 s[0:1], s[0:1]\n s_mov_b32 vcc_lo, 0x9037ab78\n v_mov_b64_e32 v[48:49], v[52:53]\n"
                      " s_mov_b32 vcc_hi, 0x3e21eeb6\n v_mov_b64_e32 v[50:51], v[54:55]\n s_xor_b64 exec, exec, s[0:1]\n", 1),
    # the same ahead of an `else` entry
    "before_else_entry": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n v_mov_b32_e32 v3, v4\n"
                          " s_mov_b64 s[4:5], s[10:11]\n s_or_saveexec_b64 s[0:1], s[0:1]\n", 1),
    # the saved mask in vcc instead of an SGPR pair
    "restore_from_vcc": ("s_and_saveexec_b64 vcc, vcc\n ds_read_b32 v1, v2\n v_mov_b32_e32 v3, v4\n"
                         " s_mov_b64 s[4:5], s[10:11]\n s_or_b64 exec, exec, vcc\n", 1),
    # other encodings of a vector copy between the scalar copy and the restore: packed, VOP3, DPP
    "packed_copy": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n s_mov_b64 s[4:5], s[10:11]\n"
                    " v_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]\n s_or_b64 exec, exec, s[0:1]\n", 1),
    "e64_copy": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n s_mov_b64 s[4:5], s[10:11]\n"
                 " v_mov_b32_e64 v3, v4\n s_or_b64 exec, exec, s[0:1]\n", 1),
    "dpp_copy": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n s_mov_b64 s[4:5], s[10:11]\n"
                 " v_mov_b32_dpp v3, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_or_b64 exec, exec, s[0:1]\n", 1),
    # an ordinary join
    "plain_join": ("s_and_saveexec_b64 s[0:1], vcc\n ds_read_b32 v1, v2\n v_add_f32_e32 v1, v1, v1\n"
                   " s_or_b64 exec, exec, s[0:1]\n", 0),
}


def _env(**extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               SF_HIP_REPORT_SGPR_SPILLS="1", SF_HIP_CACHE_DIR="off")
    for k in ("SF_HIP_UNSAFE_SGPR_SPILLS", "SF_HIP_OBJECT_DIR"):
        env.pop(k, None)
    env.update(extra)
    return env


def _plan_report(**extra):
    r = subprocess.run([sys.executable, "-c", PROGRAM], capture_output=True, text=True, env=_env(**extra), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])


@pytest.fixture(scope="module")
def base(tmp_path_factory):
    """Name and assembly of a real (tiny) generated kernel."""
    tmp = tmp_path_factory.mktemp("asm")
    info = _plan_report()
    assert info["resources"]["scratch"] == 0
    src = tmp / "k.hip"
    src.write_text(info["source"])
    asm = tmp / "k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-include",
                    "hip/hip_runtime.h", "-DSF_KERNEL_NAME=" + info["name"], "--cuda-device-only", "-S", str(src), "-o",
                    str(asm)], check=True, capture_output=True)
    return info["name"], asm.read_text(), tmp


def _object_with(base, label, snippet, sgprs):
    name, text, tmp = base
    assert text.count("s_endpgm") == 1
    text = text.replace("s_endpgm\n", "s_endpgm\n " + snippet, 1)
    unreadable = sgprs is None  # metadata the library cannot read: the register count is unknown
    sgprs = 40 if unreadable else sgprs
    text = re.sub(r"\.sgpr_count:\s+\d+", ".sgpr_count:     %d" % sgprs, text)
    text = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr %d" % (sgprs - 6), text)
    out = tmp / label
    out.mkdir(exist_ok=True)
    (out / "k.s").write_text(text)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                    str(out / "k.s"), "-o", str(out / "k.o")], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", str(out / "k.o"), "-o", str(out / (name + ".co"))], check=True)
    if unreadable:  # (the assembler insists on the key: it is renamed in the finished object)
        blob = (out / (name + ".co")).read_bytes()
        assert blob.count(b".sgpr_count") == 1
        (out / (name + ".co")).write_bytes(blob.replace(b".sgpr_count", b".sgpr_counx"))
    return str(out)


@pytest.mark.parametrize("label", sorted(PATTERNS))
def test_pattern(base, label):
    snippet, want = PATTERNS[label]
    got = _plan_report(SF_HIP_OBJECT_DIR=_object_with(base, label, snippet, 106), SF_HIP_UNSAFE_SGPR_SPILLS="1")
    assert got["name"] == base[0]
    assert got["resources"]["scratch"] // 1000 == want, (label, got["resources"])


@pytest.mark.parametrize("label", ["split_copy", "constant_select"])
def test_objects_far_from_register_exhaustion_are_not_searched(base, label):
    snippet, _ = PATTERNS[label]
    got = _plan_report(SF_HIP_OBJECT_DIR=_object_with(base, "low_pressure_" + label, snippet, 40), SF_HIP_UNSAFE_SGPR_SPILLS="1")
    assert got["resources"]["scratch"] // 1000 == 0


def test_an_object_without_readable_metadata_is_searched(base):
    """`.sgpr_count` missing: the gate cannot tell that the object is far from register
    exhaustion, so it is disassembled like a large one."""
    snippet, _ = PATTERNS["split_copy"]
    got = _plan_report(SF_HIP_OBJECT_DIR=_object_with(base, "no_metadata", snippet, None), SF_HIP_UNSAFE_SGPR_SPILLS="1")
    assert got["resources"]["scratch"] // 1000 == 1


def test_diagnostic_switches_are_named_in_the_description(base):
    snippet, _ = PATTERNS["split_copy"]
    got = _plan_report(SF_HIP_OBJECT_DIR=_object_with(base, "marked", snippet, 106), SF_HIP_UNSAFE_SGPR_SPILLS="1")
    assert "[foreign object" in got["describe"] and "[UNSAFE" in got["describe"]
    assert "[foreign" not in _plan_report()["describe"]


def test_a_flagged_object_is_replaced_by_the_next_form_of_the_kernel(base):
    """Without the diagnostic override the flagged object (here: the marching form of the generic
    kernel) is not launched; the next form of the same operator, compiled afresh, runs instead."""
    snippet, _ = PATTERNS["split_copy"]
    got = _plan_report(SF_HIP_OBJECT_DIR=_object_with(base, "refused", snippet, 106))
    assert got["all"][base[0]]["scratch"] // 1000 == 1
    launched = [ln.split()[1].rstrip(":") for ln in got["describe"].splitlines() if ln.strip().startswith("launch ")]
    assert launched and base[0] not in launched and all(got["all"][n]["scratch"] == 0 for n in launched)
