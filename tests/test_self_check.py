"""The plan-time self-check (stencilflow_amd/csrc/exec.cpp: self_check):
every fused kernel without a verdict is compared, before the plan's first use, with the
same operators run one by one by the plain generic kernel -- on the GPU, bit for bit -- and
the verdict travels with the code object through both cache levels.  The second guard
behind the EXEC-restore detector: a wrong code object is caught whatever made it wrong."""
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
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd import backend, programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
mode = sys.argv[1]
shape = (40, 24, 64)
with tempfile.TemporaryDirectory() as tmp:
    path = programs.write_program(programs.jacobi3d(shape, 4, bc_value=0.5), os.path.join(tmp, "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
out = {}
plan = Plan(sfir, options=os.environ.get("SF_TEST_OPTIONS") or None)
name = [n for n in plan.kernel_names() if n.startswith("sf_star3d")][0]
out["name"], out["source"] = name, plan.kernel_source(plan.kernel_names().index(name))
if mode == "source":
    print("RESULT " + json.dumps(out)); sys.exit(0)
lib = backend.load_library()
x = np.random.default_rng(3).uniform(-1, 1, shape).astype(np.float32)
y = np.zeros(shape, np.float32)
try:
    plan.run([x], [y])
    out["error"] = None
except Exception as exc:
    out["error"] = "{}: {}".format(type(exc).__name__, exc)
    try:
        plan.run([x], [y])
        out["second_use"] = None
    except Exception as exc2:
        out["second_use"] = str(exc2)
out["checks"] = lib.sf_self_checks_run()
out["verdicts"] = plan.kernel_verdicts()
out["nonzero"] = bool(np.abs(y).max() > 0)
plan.close()
if mode == "twice":
    backend.code_cache_stats(drop_process_level=True)   # the next plan goes to disk
    again = Plan(sfir, options=os.environ.get("SF_TEST_OPTIONS") or None)
    again.run([x], [y])
    out["checks_after_second_plan"] = lib.sf_self_checks_run()
    out["verdicts_second_plan"] = again.kernel_verdicts()
    out["disk_hits"] = backend.code_cache_stats()[0]
print("RESULT " + json.dumps(out))
"""


def _run(mode, **extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("SF_HIP_OBJECT_DIR", "SF_HIP_SELF_CHECK", "SF_HIP_UNSAFE_SGPR_SPILLS"):
        env.pop(k, None)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", PROGRAM, mode], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])


@pytest.mark.gpu
def test_a_fused_kernel_is_checked_once_and_the_verdict_travels_with_the_cache(tmp_path):
    got = _run("twice", SF_HIP_CACHE_DIR=str(tmp_path / "cache"))
    assert got["error"] is None and got["nonzero"]
    assert got["checks"] >= 1
    assert got["verdicts"][got["name"]] == 1
    # the second plan takes the object -- and its verdict -- from the disk cache: nothing is checked again
    assert got["disk_hits"] >= 1
    assert got["checks_after_second_plan"] == got["checks"]
    assert got["verdicts_second_plan"][got["name"]] == 1


@pytest.mark.gpu
def test_the_check_can_be_switched_off(tmp_path):
    got = _run("once", SF_HIP_CACHE_DIR=str(tmp_path / "cache"), SF_HIP_SELF_CHECK="0")
    assert got["error"] is None and got["checks"] == 0 and got["verdicts"][got["name"]] == 0


@pytest.mark.gpu
def test_a_wrong_code_object_is_caught_before_its_first_use(tmp_path):
    """The compiler's own assembly of the fused kernel with ONE arithmetic instruction changed
    (a double-precision add of one loop phase turned into a multiply: one element of one row of
    a quarter of the planes goes wrong), assembled and handed to the library in the compiler's
    place: the detector has nothing to say about it, the self-check refuses it.  (The tile shape is
    pinned to four thread rows so that every instruction of the step loop computes stored rows
    for some thread -- with a single thread row most rows of a tile are halo rows.)"""
    pinned = {"SF_TEST_OPTIONS": "k1.bx=64;k1.by=4;k1.rj=5"}
    info = _run("source", SF_HIP_CACHE_DIR="off", **pinned)
    name = info["name"]
    src = tmp_path / "k.hip"
    src.write_text(info["source"])
    asm = tmp_path / "k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-include",
                    "hip/hip_runtime.h", "-DSF_KERNEL_NAME=" + name, "--cuda-device-only", "-S", str(src), "-o", str(asm)],
                   check=True, capture_output=True)
    text = asm.read_text()
    assert len(re.findall(r"\bv_add_f64\b", text)) >= 5
    # one add inside the step loop (not the first ones of the prologue)
    hits = [m.start() for m in re.finditer(r"\bv_add_f64\b", text)]
    at = hits[len(hits) // 2]
    text = text[:at] + "v_mul_f64" + text[at + len("v_add_f64"):]
    out = tmp_path / "obj"
    out.mkdir()
    (out / "k.s").write_text(text)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                    str(out / "k.s"), "-o", str(out / "k.o")], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", str(out / "k.o"), "-o", str(out / (name + ".co"))], check=True)
    # unchanged, the hand-assembled object passes (the path itself is sound) ...
    good = tmp_path / "good"
    good.mkdir()
    (good / "k.s").write_text(asm.read_text())
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                    str(good / "k.s"), "-o", str(good / "k.o")], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", str(good / "k.o"), "-o", str(good / (name + ".co"))], check=True)
    ok = _run("once", SF_HIP_CACHE_DIR="off", SF_HIP_OBJECT_DIR=str(good), **pinned)
    assert ok["error"] is None and ok["verdicts"][name] == 1
    # ... with the changed instruction it is refused, and stays refused
    bad = _run("once", SF_HIP_CACHE_DIR="off", SF_HIP_OBJECT_DIR=str(out), **pinned)
    assert bad["error"] is not None and "self-check" in bad["error"], bad
    assert bad["error"].startswith("ValueError") and bad["verdicts"][name] == 2
    assert bad["second_use"] is not None and "self-check" in bad["second_use"]
