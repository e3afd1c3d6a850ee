#!/usr/bin/env python3
"""Instruction budget of the kernels a plan runs: the PRODUCT's code objects (what hipRTC
compiled with the plan's own flags, fetched through sf_plan_kernel_object) are disassembled
with llvm-objdump, the steady-state loop of each fused kernel is located (the backward branch
spanning the most instructions) and its instructions are counted by class, per cell update:

    arith   v_add / v_mul / v_fma / v_sub ... on f32 / f64 (packed forms count per instruction)
    cvt     v_cvt_*                      dpp+mov  v_mov (DPP or not), v_pk_mov, v_accvgpr_*, v_readlane ...
    select  v_cndmask, v_cmp*            lds      ds_*            vmem  buffer_* / global_* / scratch_*
    salu    s_* other than waits         wait     s_waitcnt, s_nop, s_barrier, s_sleep

A loop iteration of a plane-streaming kernel is `unroll` steps of `T` operators on `RJ x VK`
points per thread (macros of the generated source), so updates per thread and iteration =
unroll * T * RJ * VK; instructions / update is the static count of the loop divided by that --
both arms of a wave-uniform branch are counted, so boundary-only code makes the figure an upper
bound.  Works without a GPU (plan creation compiles for gfx950).

usage: isa_stats.py program.json [options] [--keep DIR] [--json]
       isa_stats.py --families [--json]      the benchmark workloads of every fused kernel family"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CLASSES = ("arith", "cvt", "dpp+mov", "select", "other_valu", "lds", "vmem", "salu", "wait")


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith(("v_mov", "v_pk_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_swap", "v_perm")):
        return "dpp+mov"
    if op.startswith(("v_cndmask", "v_cmp")):
        return "select"
    if re.match(r"v_(pk_)?(add|sub|mul|fma|mac|fmac|max|min|rcp|sqrt|rsq|div|exp|log|sin|cos|ldexp|frexp|trig|floor|ceil|rndne|trunc|fract)", op):
        return "arith"
    return "other_valu"


def disassemble(code):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code)
        f.flush()
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True, text=True).stdout
    insts = []  # (address, mnemonic, operands)
    for line in out.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return insts


def main_loop(insts):
    """(first, last) instruction indices of the backward branch that spans the most instructions."""
    addr_ix = {a: i for i, (a, _, _) in enumerate(insts)}
    best = None
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"(-?\d+)\s*$", args)
        if not m:
            continue
        simm = int(m.group(1))
        if simm >= 32768:  # llvm-objdump prints the 16-bit field unsigned
            simm -= 65536
        target = a + 4 + 4 * simm  # SOPP branch: PC + 4 + simm16 * 4
        if target in addr_ix and addr_ix[target] < i:
            span = i - addr_ix[target]
            if best is None or span > best[1] - best[0]:
                best = (addr_ix[target], i)
    return best


def macros(source):
    d = {}
    for m in re.finditer(r"^#define (SF_[A-Z0-9_]+) (-?\d+)\s*$", source, re.M):
        d.setdefault(m.group(1), int(m.group(2)))  # (codegen's definitions come first; the skeleton may redefine)
    return d


def updates_per_iteration(name, mac):
    """Cell updates per thread in one iteration of the step loop."""
    t, rj, vk = mac.get("SF_T", 1), mac.get("SF_RJ", 1), mac.get("SF_VK", 1)
    if name.startswith("sf_star"):
        unroll = 3 + (1 if mac.get("SF_RING4", 0) else 0)  # (the four-slot input ring: the step loop is unrolled by four)
    elif name.startswith("sf_compact"):
        unroll = 4
    elif name.startswith("sf_wstar"):
        unroll = 5
    elif name.startswith("sf_dense"):
        unroll, t = (mac.get("SF_ACCS", 5) if mac.get("SF_DENSE_STREAM", 0) else 6), 1  # (the streaming form rotates its accumulator sets)
        if mac.get("SF_DENSE_T2", 0):
            # (two or three fused operators, 2 RS + 1 accumulator sets each: the step loop is unrolled by that many)
            unroll, t = mac.get("SF_ACCS", 3), mac.get("SF_NST", 2)
        vk = mac.get("SF_VK", 4)
    else:
        return None
    return unroll * t * rj * vk


def budget(plan, index, keep=None):
    name = plan.kernel_names()[index]
    code, flags = plan.kernel_object(index)
    if keep:
        open(os.path.join(keep, name + ".co"), "wb").write(code)
        open(os.path.join(keep, name + ".hip"), "w").write(plan.kernel_source(index))
    insts = disassemble(code)
    rec = {"kernel": name, "flags": flags, "instructions": len(insts)}
    rec.update(plan.kernel_resources()[name])
    loop = main_loop(insts)
    upd = updates_per_iteration(name, macros(plan.kernel_source(index)))
    counts = collections.Counter()
    ops = collections.Counter()
    body = insts[loop[0]:loop[1] + 1] if loop else insts
    for _, op, _ in body:
        counts[classify(op)] += 1
        ops[op] += 1
    rec["loop_instructions"] = len(body)
    rec["updates_per_thread_and_iteration"] = upd
    rec["by_class"] = {c: counts.get(c, 0) for c in CLASSES}
    if upd:
        rec["per_update"] = {c: round(counts.get(c, 0) / upd, 2) for c in CLASSES}
        rec["per_update"]["all_valu"] = round(sum(counts.get(c, 0) for c in ("arith", "cvt", "dpp+mov", "select", "other_valu")) / upd, 2)
        rec["per_update"]["total"] = round(len(body) / upd, 2)
    rec["top"] = ops.most_common(10)
    return rec


def show(rec):
    print("{kernel}  flags '{flags}'  vgpr {vgprs} lds {lds}".format(**rec))
    print("   loop: {} instructions, {} updates per thread and iteration".format(rec["loop_instructions"],
                                                                              rec["updates_per_thread_and_iteration"]))
    if "per_update" in rec:
        pu = rec["per_update"]
        print("   per update: " + "  ".join("{} {}".format(c, pu[c]) for c in CLASSES) +
              "  | VALU {}  total {}".format(pu["all_valu"], pu["total"]))
    print("   top: " + ", ".join("{} {}".format(o, n) for o, n in rec["top"]))


FAMILIES = {
    "star (C3 jacobi3d 512^3 f32)": (lambda: programs.jacobi3d((512, 512, 512), 8), None),
    "star 2-D (C2 jacobi2d 4096^2 f32)": (lambda: programs.jacobi2d((4096, 4096), 8), None),
    "star f64 (C5 chain 512^3)": (lambda: programs.diffusion_advection_laplacian((512, 512, 512)), None),
    "dense, two fused (27-point box 512^3 f32)": (lambda: programs.synthesize("float32", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")[0], None),
    "compact (27-point box 512^3 f32, dense.t2=0)": (lambda: programs.synthesize("float32", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")[0], "dense.t2=0"),
    "wide star (radius-2 cross 512^3 f32)": (lambda: programs.synthesize("float32", 4, 0.0, 512, 512, 512, 2, 2, 2)[0], None),
    "dense (125-point box 512^3 f32)": (lambda: programs.synthesize("float32", 2, 0.0, 512, 512, 512, 2, 2, 2, stencil_shape="box")[0], None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("program", nargs="?")
    ap.add_argument("options", nargs="?", default=None)
    ap.add_argument("--families", action="store_true")
    ap.add_argument("--keep", default=os.environ.get("SF_KEEP"))
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        if args.families:
            for label, (make, opts) in FAMILIES.items():
                jobs.append((label, programs.write_program(make(), os.path.join(tmp, "p%d.json" % len(jobs))), opts))
        else:
            jobs.append((args.program, args.program, args.options))
        for label, path, opts in jobs:
            plan = Plan(lower(sf.KernelChainGraph(path)), options=opts)
            seen = set()
            for i, name in enumerate(plan.kernel_names()):
                if name in seen or name.startswith("sf_point"):
                    continue
                seen.add(name)
                rec = budget(plan, i, args.keep)
                rec["workload"] = label
                if args.json:
                    print(json.dumps(rec))
                else:
                    print("# " + label)
                    show(rec)
            plan.close()


if __name__ == "__main__":
    main()
