"""The code objects the library refuses to run (stencilflow_amd/csrc/codecache.cpp:
count_late_exec_restores): on this toolchain a register-allocator
copy can end up ahead of the EXEC restore of a join block and then runs for the
lanes of the `if` body only.  The detector reads the machine code of every compiled
kernel; a flagged object is never launched (the fused group is shortened instead).
Compile-only: no GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys, tempfile
import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
out = {}
with tempfile.TemporaryDirectory() as tmp:
    # a pinned tile shape whose three-operator object is known to carry the fault (tools/spill_probe2.py)
    path = programs.write_program(programs.jacobi3d((14, 30, 64), 3, bc_value=0.25), os.path.join(tmp, "p.json"))
    plan = Plan(lower(sf.KernelChainGraph(path)), options={"fuse": 3, "k1.bx": 128, "k1.by": 2, "k1.rj": 8, "allow_spills": 1})
    out["pinned"] = {"describe": plan.describe(), "resources": plan.kernel_resources()}
    plan.close()
    # the benchmark's kernel
    path = programs.write_program(programs.jacobi3d((512, 512, 512), 2), os.path.join(tmp, "q.json"))
    plan = Plan(lower(sf.KernelChainGraph(path)))
    out["c3"] = {"describe": plan.describe(), "resources": plan.kernel_resources()}
    plan.close()
print("RESULT " + json.dumps(out))
"""


def _run(extra_env):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               SF_HIP_REPORT_SGPR_SPILLS="1", SF_HIP_CACHE_DIR="off")
    env.pop("SF_HIP_UNSAFE_SGPR_SPILLS", None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])


def _launched(entry):
    return [ln.split()[1].rstrip(":") for ln in entry["describe"].splitlines() if ln.strip().startswith("launch ")]


def test_flagged_object_is_compiled_but_never_launched():
    got = _run({})
    pinned = got["pinned"]
    # (with SF_HIP_REPORT_SGPR_SPILLS `scratch` = SGPR spills + 1000 x flagged EXEC restores)
    flagged = [n for n, r in pinned["resources"].items() if r["scratch"] >= 1000]
    assert flagged and all("_t3_" in n for n in flagged), pinned["resources"]
    launched = _launched(pinned)
    assert launched and not set(flagged) & set(launched), pinned["describe"]
    # the three operators still run: as a group of two and a single one
    assert sorted(n.split("_")[3] for n in launched) == ["t1", "t2"]
    # the benchmark's kernel: no scalar spills, nothing flagged
    c3 = got["c3"]
    (name, res), = c3["resources"].items()
    assert name.startswith("sf_star3d_f32_t2_") and res["scratch"] == 0 and res["spills"] == 0 and res["agprs"] == 0


def test_diagnostic_override_runs_the_flagged_object():
    got = _run({"SF_HIP_UNSAFE_SGPR_SPILLS": "1"})
    launched = _launched(got["pinned"])
    assert len(launched) == 1 and "_t3_" in launched[0]
    assert got["pinned"]["resources"][launched[0]]["scratch"] >= 1000
