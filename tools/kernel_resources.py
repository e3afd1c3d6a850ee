#!/usr/bin/env python3
"""Compile the kernels a plan generates offline with hipcc and print their
register / LDS / spill figures (-Rpass-analysis=kernel-resource-usage).
Works without a GPU.  usage: kernel_resources.py program.json [opt=val;...]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402


def resources(program_path, options=None, keep=None):
    chain = sf.KernelChainGraph(program_path)
    plan = Plan(lower(chain), options=options)
    out = {}
    for i, name in enumerate(plan.kernel_names()):
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(keep or tmp, name + ".hip")
            with open(src, "w") as f:
                f.write(plan.kernel_source(i))
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3",
                   "-std=c++17", "-ffp-contract=off", "-include",
                   "hip/hip_runtime.h", "-DSF_KERNEL_NAME=" + name, "-c", src,
                   "-o", os.path.join(tmp, "k.o"),
                   "-Rpass-analysis=kernel-resource-usage"]
            if keep:
                cmd += ["-save-temps=obj"]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=keep or tmp)
            info = {}
            for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]",
                        "Occupancy \\[waves/SIMD\\]", "VGPRs Spill",
                        "LDS Size \\[bytes/block\\]", "TotalSGPRs"):
                m = re.search(r"\s" + key + r": (\d+)", r.stderr)
                if m:
                    info[key.replace("\\", "")] = int(m.group(1))
            if r.returncode != 0:
                info["error"] = r.stderr[-2000:]
            out[name] = info
    return plan.describe().splitlines()[1] if plan.num_launches else "", out


if __name__ == "__main__":
    opts = sys.argv[2] if len(sys.argv) > 2 else None
    d, res = resources(sys.argv[1], opts, keep=os.environ.get("SF_KEEP"))
    print(d)
    for k, v in res.items():
        print(k, v)
