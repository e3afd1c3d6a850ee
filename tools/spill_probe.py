#!/usr/bin/env python3
"""GPU: hunt for the wrong results of star-kernel code objects that spill
(profiles/r01_config_fuzz.log: every failing pinned shape reports spills, scratch
or AGPRs).  Runs jacobi3d 14x30x64 x 4 operators on pinned shapes that are likely
to spill, prints for every failing shape where the result differs, and keeps the
generated source of failing and passing shapes under gpurun_out/spill/ for an
offline look at the ISA (tools/isa_stats.py works on a program; here the source
is saved as the library compiled it).
usage: SF_HIP_UNSAFE_SGPR_SPILLS=1 SF_HIP_REPORT_SGPR_SPILLS=1 spill_probe.py [extra options, e.g. "fuse=3"]
(the first variable lets the library run code objects that spill SGPRs -- the
thing under test; the second makes `scratch` of the resource record carry the
SGPR spill count)"""
import itertools
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402
from oracle import numpy_oracle as npo  # noqa: E402


def main():
    extra = {}
    if len(sys.argv) > 1:
        extra = {k: v for k, v in (kv.split("=") for kv in sys.argv[1].split(";") if kv)}
    rng = np.random.default_rng(99)
    shape, stages = (14, 30, 64), 4
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    out_dir = os.path.join("gpurun_out", "spill")
    os.makedirs(out_dir, exist_ok=True)
    kept = {"fail": 0, "pass_spill": 0}
    tally = {"shapes": 0, "failing": 0, "flagged": 0, "failing_unflagged": 0, "flagged_passing": 0,
             "sgpr_spilling_unflagged_passing": 0}
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
        for fuse, bx, by, rj in itertools.product([2, 3], [64, 128], [1, 2, 4, 8], [5, 6, 7, 8]):
            if bx * by > 1024 or by * rj - 2 * fuse < 1:
                continue
            opt = dict({"fuse": fuse, "k1.bx": bx, "k1.by": by, "k1.rj": rj, "allow_spills": 1}, **extra)
            try:
                plan = Plan(sfir, options=opt)
            except ValueError:
                continue
            res = plan.kernel_resources()
            got = np.zeros_like(want)
            plan.run([x], [got], 1)
            bad = np.argwhere(got != want)
            spilled = any(r["spills"] or r["scratch"] or r["agprs"] for r in res.values())
            # (SF_HIP_REPORT_SGPR_SPILLS=1: `scratch` = SGPR spills + 1000 x EXEC restores behind allocator code)
            late = sum(r["scratch"] // 1000 for r in res.values())
            line = {"opt": opt, "res": list(res.values()), "spilled": spilled, "bad_points": int(len(bad)),
                    "sgpr_spills": sum(r["scratch"] % 1000 for r in res.values()), "late_exec_restores": late}
            tally["shapes"] += 1
            tally["failing"] += bool(len(bad))
            tally["flagged"] += bool(late)
            tally["failing_unflagged"] += bool(len(bad)) and not late
            tally["flagged_passing"] += bool(late) and not len(bad)
            tally["sgpr_spilling_unflagged_passing"] += (not late) and (not len(bad)) and line["sgpr_spills"] > 0
            if len(bad):
                line["bad_i"] = sorted(set(int(b[0]) for b in bad))[:20]
                line["bad_j"] = sorted(set(int(b[1]) for b in bad))[:40]
                line["bad_k"] = sorted(set(int(b[2]) for b in bad))[:70]
                line["maxrel"] = npo.max_rel_err(want, got)
                line["nan"] = int(np.isnan(got).sum())
            print(json.dumps(line), flush=True)
            tag = "fail" if len(bad) else ("pass_spill" if spilled else None)
            if tag and kept[tag] < 3:
                kept[tag] += 1
                for i, name in enumerate(plan.kernel_names()):
                    with open(os.path.join(out_dir, "{}_{}_{}.hip".format(tag, kept[tag], name)), "w") as f:
                        f.write("// " + json.dumps(line) + "\n" + plan.kernel_source(i))
            plan.close()
    print("# " + json.dumps(tally), flush=True)


if __name__ == "__main__":
    main()
