#!/usr/bin/env python3
"""CPU: a digest of the machine code (.text of the gfx950 code object) of every kernel a set of reference plans compiles
-- the check behind edits of the kernel skeletons that must not change what the product runs: a kernel's NAME carries
a hash of its generated source (skeleton text included), so the names move with every edit of a comment; the
instructions must not.
usage: object_digest.py [--out FILE] [--only c3,c2,...]     (needs no GPU; SF_HIP_LIBNAME picks another build)
       object_digest.py --compare A.json B.json"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def text_digest(code):
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "k.co"), os.path.join(tmp, "k.text")
        with open(src, "wb") as f:
            f.write(code)
        subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.text", src, dst])
        with open(dst, "rb") as f:
            text = f.read()
    return hashlib.sha256(text).hexdigest()[:16], len(text)


def _not_a_plain_sum(prog):
    """the generator's 125-point box with its first `+` turned into `-`: the dense kernel's general form (the text as it stands)"""
    for k in prog["program"].values():
        k["computation_string"] = k["computation_string"].replace(" + ", " - ", 1)
    return prog


def cases():
    from stencilflow_amd import programs
    syn = programs.synthesize
    return {
        "c3": (programs.jacobi3d((512, 512, 512), 8), "dense.t2=0"),
        "c3_fused": (programs.jacobi3d((512, 512, 512), 9), None),
        "c3_t1": (programs.jacobi3d((512, 512, 512), 3), "fuse=1"),
        "c3_slab": (programs.jacobi3d((512, 512, 512), 8), "slab=0:512:16:4096;dense.t2=0"),
        "c2": (programs.jacobi2d((4096, 4096), 8), "fuse=4"),
        "c5": (programs.diffusion_advection_laplacian((512, 512, 512)), "fuse=3"),
        "generic": (programs.jacobi3d((64, 64, 64), 2), "generic_only=1"),
        "box_compact": (syn("float32", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")[0], "dense.t2=0"),
        "extra_fields": (syn("float32", 4, 0.5, 512, 512, 512, 1, 1, 1)[0], None),
        "box2d_compact": (syn("float32", 4, 0.0, 4096, 4096, 0, 1, 1, 0, stencil_shape="box")[0], "dense.t2=0"),
        "hotspot": (syn("float32", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="hotspot")[0], None),
        "hotspot2d": (syn("float32", 4, 0.0, 4096, 4096, 0, 1, 1, 0, stencil_shape="hotspot")[0], None),
        "cross_f64": (syn("float64", 4, 0.0, 512, 512, 512, 1, 1, 1)[0], None),
        "wide": (syn("float32", 4, 0.0, 512, 512, 512, 2, 2, 2)[0], "dense.t2=0"),
        "cross2_fused": (syn("float32", 4, 0.0, 512, 512, 512, 2, 2, 2)[0], None),
        "box_fused": (syn("float32", 4, 0.0, 512, 512, 512, 1, 1, 1, stencil_shape="box")[0], None),
        "dense": (syn("float32", 2, 0.0, 512, 512, 512, 2, 2, 2, stencil_shape="box")[0], None),
        "cross3": (syn("float32", 2, 0.0, 512, 512, 512, 3, 3, 3)[0], None),
        "wide_f64": (syn("float64", 2, 0.0, 512, 512, 512, 2, 2, 2)[0], None),
        "wide2d": (syn("float32", 4, 0.0, 4096, 4096, 0, 2, 2, 0)[0], None),
        "fork": (syn("float32", 12, 0.0, 512, 512, 512, 1, 1, 1, fork_frequency=0.25)[0], None),
        "fork2d": (syn("float32", 12, 0.0, 4096, 4096, 0, 1, 1, 0, fork_frequency=0.25)[0], None),
        "fork_f64": (syn("float64", 12, 0.0, 512, 512, 512, 1, 1, 1, fork_frequency=0.25)[0], None),
        "dense_general": (_not_a_plain_sum(syn("float32", 2, 0.0, 512, 512, 512, 2, 2, 2, stencil_shape="box")[0]), None),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--compare", nargs=2)
    args = ap.parse_args()
    if args.compare:
        a, b = (json.load(open(p)) for p in args.compare)
        bad = 0
        for case in sorted(set(a) | set(b)):
            da = sorted(v for v in a.get(case, {}).values())
            db = sorted(v for v in b.get(case, {}).values())
            same = da == db
            bad += not same
            print(("same     " if same else "DIFFERENT") + " " + case + ("" if same else "  %s -> %s" % (da, db)))
        print("cases that differ:", bad)
        return 1 if bad else 0
    import stencilflow_amd as sf
    from stencilflow_amd import backend, programs
    from stencilflow_amd.lowering import lower
    only = [c for c in args.only.split(",") if c]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (prog, opts) in cases().items():
            if only and name not in only:
                continue
            path = programs.write_program(prog, os.path.join(tmp, name + ".json"))
            with backend.Plan(lower(sf.KernelChainGraph(path)), options=opts) as plan:
                out[name] = {}
                for i, kname in enumerate(plan.kernel_names()):
                    code, _flags = plan.kernel_object(i)
                    family = kname.rsplit("_", 1)[0]
                    out[name]["%s#%d" % (family, i)] = list(text_digest(code))
            print(name, out[name], flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
