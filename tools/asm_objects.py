#!/usr/bin/env python3
"""CPU: hand-assembled code objects for the shapes of tools/spill_probe2.py -- the
compiler's own assembly (hipcc -S, the flags of the library's hipRTC call), as it
is or with instructions padded, assembled and linked into
<out>/<variant>/<kernel name>.co.  On the GPU box SF_HIP_OBJECT_DIR=<out>/<variant>
makes the library run these instead of compiling (csrc/codecache.cpp: intern_kernel).
Purpose: tell a hazard / timing problem (goes away with padding) from a logical
miscompile (stays) in the code objects that spill SGPRs, DESIGN.md §5.1.
usage: asm_objects.py <out dir> [variant ...]     variants: asis nop_all nop_sgprw nop_vmem nop_lane,
or "name=<compiler flags>": the compiler's output under further flags, e.g. "nocp=-mllvm -disable-copyprop"."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SF_HIP_UNSAFE_SGPR_SPILLS"] = "1"
os.environ.setdefault("SF_HIP_CACHE_DIR", "off")
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402
from tools.spill_probe2 import CONTROL, FAILING  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
INSTR = re.compile(r"^\s+(v_|s_|ds_|buffer_|global_|flat_|scratch_)")
# VALU instructions with a scalar destination
SGPR_WRITER = re.compile(r"^\s+(v_readlane_b32|v_readfirstlane_b32|v_cmpx?_\w+_e64|v_add_co_u32|v_addc_co_u32|"
                         r"v_sub_co_u32|v_subb_co_u32|v_div_scale_\w+|v_mad_[ui]64_[ui]32)\s+(s\[?\d+|vcc)")
PAD = "\ts_nop 7\n"


def hoist_exec_restores(lines):
    """The repair under test: an `s_or_b64 exec, exec, s[a:b]` that the compiler left BEHIND copies at
    the top of its block (register-allocator split copies and spill code, which then run under the
    narrowed EXEC of the preceding masked block) moves to the top of the block."""
    out, moved = list(lines), 0
    for i, line in enumerate(lines):
        m = re.match(r"^\s+s_or_b64 exec, exec, s\[(\d+):(\d+)\]", line)
        if not m:
            continue
        saved = {int(m.group(1)), int(m.group(2))}
        j = i - 1
        while j >= 0 and INSTR.match(out[j]) and re.match(
                r"^\s+(v_accvgpr_(write|read|mov)_b32|v_mov_b(32|64)_e32|s_mov_b(32|64)|v_readlane_b32|v_writelane_b32|"
                r"scratch_(load|store)_\w+|s_nop)\s", out[j]) and "exec" not in out[j]:
            d = re.match(r"^\s+(s_mov_b64 s\[(\d+):|s_mov_b32 s(\d+)|v_readlane_b32 s(\d+))", out[j])
            if d and int([g for g in d.groups()[1:] if g][0]) in saved | {min(saved) - 1}:
                break
            j -= 1
        # j: last line that is not a movable intruder; hoist only to a block label
        if j < i - 1 and re.match(r"^(; %bb\.\d+:|\.LBB\d+_\d+:)", out[j]):
            out.insert(j + 1, out.pop(i))
            moved += 1
    return out, moved


def transform(lines, variant):
    if variant == "nop_hoist" or variant == "hoist":
        out, moved = hoist_exec_restores(lines)
        transform.moved = moved
        return out
    out = []
    for line in lines:
        is_instr = bool(INSTR.match(line)) and not line.lstrip().startswith("s_code_end")
        if variant == "nop_all" and is_instr:
            out.append(PAD)
        if variant == "nop_vmem" and re.match(r"^\s+buffer_", line):
            out.append(PAD)
        if variant == "nop_lane" and re.match(r"^\s+v_(readlane|writelane)_b32", line):
            out.append(PAD)
        out.append(line)
        if variant == "nop_sgprw" and SGPR_WRITER.match(line):
            out.append(PAD)
        if variant == "nop_lane" and re.match(r"^\s+v_(readlane|writelane)_b32", line):
            out.append(PAD)
    return out


def build_one(job):
    out_dir, tmp, name, variants = job
    src = os.path.join(tmp, name + ".hip")
    meta = ""
    for variant in variants:
        label, _, flags = variant.partition("=")
        asm = os.path.join(tmp, name + "." + label + ".s")
        if label.startswith("uncond"):
            # source-level experiment: the two LDS reads of the neighbouring thread rows' edge rows
            # become unconditional (clamped thread row) -- no definition under a narrowed EXEC
            text = open(src).read()
            a = "if (ty > 0)\n      jm0 = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty - 1, 1) + tx * SF_VK]);"
            b = "if (ty < SF_BY - 1)\n      jpl = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty + 1, 0) + tx * SF_VK]);"
            if a not in text or b not in text:
                raise SystemExit("variant uncond: the kernel skeleton already reads these rows unconditionally "
                                 "(stencilflow_amd/csrc/kernels/star3d.h since the end of round 2)")
            text = text.replace(a, "jm0 = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty > 0 ? ty - 1 : 0, 1) + tx * SF_VK]);")
            text = text.replace(b, "jpl = *reinterpret_cast<const sf_vec*>(&lds[sf_rows_at(src, ty < SF_BY - 1 ? ty + 1 : ty, 0) + tx * SF_VK]);")
            src_used = os.path.join(tmp, name + "." + label + ".hip")
            with open(src_used, "w") as f:
                f.write(text)
        else:
            src_used = src
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-include", "hip/hip_runtime.h", "-DSF_KERNEL_NAME=" + name, "--cuda-device-only"] +
                       flags.split() + ["-S", src_used, "-o", asm], check=True, capture_output=True)
        lines = open(asm).readlines()
        meta += " %s:%s" % (label, ",".join(ln.split()[-1] for ln in lines if "sgpr_spill_count" in ln or
                                            ".vgpr_spill_count" in ln))
        os.makedirs(os.path.join(out_dir, label), exist_ok=True)
        if not flags and (label.startswith("nop_") or label == "hoist"):
            with open(asm, "w") as f:
                f.writelines(transform(lines, label))
            if label == "hoist":
                meta += "(%d exec restores hoisted)" % transform.moved
        obj = os.path.join(tmp, name + "." + label + ".o")
        subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950",
                        "-c", asm, "-o", obj], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", obj, "-o", os.path.join(out_dir, label, name + ".co")], check=True)
        if os.environ.get("SF_KEEP_ASM"):
            os.makedirs(os.environ["SF_KEEP_ASM"], exist_ok=True)
            os.replace(asm, os.path.join(os.environ["SF_KEEP_ASM"], name + "." + label + ".s"))
    return name, meta


def main():
    from concurrent.futures import ThreadPoolExecutor
    out_dir = sys.argv[1]
    variants = sys.argv[2:] or ["asis", "nop_all", "nop_sgprw"]
    prog = programs.jacobi3d((14, 30, 64), 3, bc_value=0.25)
    with tempfile.TemporaryDirectory() as tmp:
        path = programs.write_program(prog, os.path.join(tmp, "p.json"))
        sfir = lower(sf.KernelChainGraph(path))
        jobs, shapes = [], {}
        for bx, by, rj in FAILING + CONTROL:
            plan = Plan(sfir, options={"fuse": 3, "k1.bx": bx, "k1.by": by, "k1.rj": rj, "allow_spills": 1})
            for i, name in enumerate(plan.kernel_names()):
                with open(os.path.join(tmp, name + ".hip"), "w") as f:
                    f.write(plan.kernel_source(i))
                jobs.append((out_dir, tmp, name, variants))
                shapes[name] = (bx, by, rj)
            plan.close()
        with ThreadPoolExecutor(max_workers=7) as pool:
            for name, meta in pool.map(build_one, jobs):
                print("%3dx%d rj %d  %s  (sgpr,vgpr spills)%s" % (shapes[name] + (name, meta)), flush=True)


if __name__ == "__main__":
    main()
