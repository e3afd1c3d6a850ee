#!/usr/bin/env python3
"""CPU: TIMING-ONLY variants of a dense-kernel code object (parts of the step switched off; the results are wrong by
construction), built outside the library: the plan's generated source is edited, compiled with hipcc under the flags of
the library's hipRTC call and linked into <out>/<variant>/<kernel name>.co.  On the GPU box
SF_HIP_OBJECT_DIR=<out>/<variant> SF_HIP_SELF_CHECK=0 makes the library launch that object in place of its own
(csrc/codecache.cpp: intern_kernel) -- e.g. under tools/dense_probe.py --no-check.  This replaces round 4's
`debug.whatif` plan option: no wrong-result build is reachable through sf_plan_create any more.
usage: whatif_objects.py WORKLOAD "PLAN OPTIONS" OUT_DIR [variant ...]
  variants: asis nobar nolds nodma nostore nomem (= nodma + nostore) valu (= everything but the arithmetic) ahead2 carry"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SF_HIP_CACHE_DIR", "off")
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402
from tools.dense_probe import WORKLOADS  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
FAKE = ("template <typename V> static __device__ __forceinline__ V sf_fake() { V v; asm volatile(\"\" : \"=v\"(v)); return v; }\n")


def edit(text, variant):
    parts = {"nomem": ("nodma", "nostore"), "valu": ("nobar", "nolds", "nodma", "nostore")}.get(variant, (variant,))
    for part in parts:
        if part == "asis":
            continue
        elif part in ("ahead2", "ahead3"):  # (more input slots: planes requested two / three steps ahead -- the results stay right)
            text = re.sub(r"#define SF_IN_SLOTS (\d+)", lambda m: "#define SF_IN_SLOTS %d" % (int(m.group(1)) + int(part[-1]) - 1), text)
        elif part == "carry":
            # what keeping the converted own columns of a plane for the step that reads its centre row again would save
            # (radius-1 crosses in the fused forms: the middle four of the six values of segment g1_<centre row>)
            text = text.replace("typedef float sf_t;", "typedef float sf_t;\nstatic __device__ __forceinline__ double sf_fake_d() { double v; asm volatile(\"\" : \"=v\"(v)); return v; }\n", 1)
            text = re.sub(r"\(double\)g1_3\[[1-4]\]", "sf_fake_d()", text)
        elif part == "nobar":
            text = text.replace("\\n\\ts_barrier", "").replace('asm volatile("s_barrier" ::: "memory");', "")
        elif part == "nolds":
            text = text.replace("typedef float sf_t;", "typedef float sf_t;\n" + FAKE, 1).replace("typedef double sf_t;", "typedef double sf_t;\n" + FAKE, 1)
            text = re.sub(r"\*reinterpret_cast<const sfd_chunk\*>\(src \+ \d+\)", "sf_fake<sfd_chunk>()", text)
            text = re.sub(r"\*reinterpret_cast<const sfd_pair\*>\(src \+ \d+\)", "sf_fake<sfd_pair>()", text)
            text = re.sub(r"= src\[\d+\];", "= sf_fake<sf_t>();", text)
        elif part == "nodma":
            text = text.replace("buffer_load_dwordx4 %1, %2, 0 offen lds", "s_nop 0")
        elif part == "nostore":
            text = text.replace("sf_buf_store<sf_vec, (SF_NT & 1) ? 2 : 0>(o, rs, cx.st_off[r]);",
                                "asm volatile(\"\" : : \"v\"(o), \"s\"(rs), \"v\"(cx.st_off[r]));")
        else:
            raise SystemExit("unknown variant part " + part)
    return text


def main():
    workload, opts, out_dir = sys.argv[1], sys.argv[2] or None, sys.argv[3]
    variants = sys.argv[4:] or ["asis", "nobar", "nolds", "nomem", "valu"]
    dtype, dims, extent, stages = WORKLOADS[workload][:4]
    ext = [extent if d else 0 for d in dims]
    shape = (WORKLOADS[workload] + ("box",))[4]
    prog = programs.jacobi3d(tuple(dims), stages) if shape == "jacobi3d" else programs.synthesize(dtype, stages, 0.0, *dims, *ext, stencil_shape=shape)[0]
    with tempfile.TemporaryDirectory() as tmp:
        plan = Plan(lower(sf.KernelChainGraph(programs.write_program(prog, os.path.join(tmp, "p.json")))), options=opts)
        name, source = plan.kernel_names()[0], plan.kernel_source(0)
        _code, flags = plan.kernel_object(0)
        plan.close()
        for variant in variants:
            src = os.path.join(tmp, variant + ".hip")
            with open(src, "w") as f:
                f.write(edit(source, variant))
            asm, obj = os.path.join(tmp, variant + ".s"), os.path.join(tmp, variant + ".o")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-include",
                            "hip/hip_runtime.h", "-DSF_KERNEL_NAME=" + name, "--cuda-device-only", "-Wno-everything"] + flags.split() +
                           ["-S", src, "-o", asm], check=True)
            subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", obj], check=True)
            os.makedirs(os.path.join(out_dir, variant), exist_ok=True)
            subprocess.run([LLVM + "/ld.lld", "-shared", obj, "-o", os.path.join(out_dir, variant, name + ".co")], check=True)
            text = open(asm).read()
            print(variant, name, "v_add_f32", len(re.findall(r"^\s+v_add_f32", text, re.M)), "ds_read", len(re.findall(r"^\s+ds_read", text, re.M)),
                  "barriers", len(re.findall(r"^\s+s_barrier", text, re.M)), "vgprs", re.findall(r"\.vgpr_count:\s+(\d+)", text), flush=True)


if __name__ == "__main__":
    main()
