#!/usr/bin/env python3
"""Instruction census of the kernels a plan generates (offline, hipcc -S).
usage: isa_stats.py program.json [options]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


def main():
    chain = sf.KernelChainGraph(sys.argv[1])
    plan = Plan(lower(chain), options=sys.argv[2] if len(sys.argv) > 2 else None)
    keep = os.environ.get("SF_KEEP")
    for i, name in enumerate(plan.kernel_names()):
        with tempfile.TemporaryDirectory() as tmp:
            d = keep or tmp
            src = os.path.join(d, name + ".hip")
            open(src, "w").write(plan.kernel_source(i))
            asm = os.path.join(d, name + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                            "-ffp-contract=off", "-include", "hip/hip_runtime.h",
                            "-DSF_KERNEL_NAME=" + name, "--cuda-device-only", "-S", src, "-o", asm],
                           check=True, capture_output=True)
            c = collections.Counter()
            for line in open(asm):
                t = line.strip().split()
                if t and re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_)", t[0]):
                    c[t[0]] += 1
            total = sum(c.values())
            groups = collections.Counter()
            for op, n in c.items():
                if op.startswith("v_mov"):
                    groups["v_mov"] += n
                elif op.endswith("f64") or "f64" in op:
                    groups["f64"] += n
                elif op.startswith("v_"):
                    groups["valu_other"] += n
                elif op.startswith("ds_"):
                    groups["lds"] += n
                elif op.startswith(("global_", "buffer_", "scratch_")):
                    groups["vmem"] += n
                elif op.startswith("s_waitcnt"):
                    groups["waitcnt"] += n
                else:
                    groups["salu"] += n
            print(name, "total", total, dict(groups))
            print("   top:", c.most_common(14))


if __name__ == "__main__":
    main()
