"""Seeded generator of random stencil programs in the reference's JSON format,
used to compare implementations on inputs nobody hand-picked: random DAGs,
offsets (also farther than the domain), mixed float32/float64, lower-
dimensional and scalar inputs, program constants, int and float boundary
literals, `shrink`, ternaries / comparisons / boolean operators, min / max,
multi-statement kernels.  Only + - * (no division, no transcendental calls), so
every implementation must agree bit for bit."""
import numpy as np

ITERATORS = ["i", "j", "k"]


def _offset_text(it, o):
    return it if o == 0 else ("{}+{}".format(it, o) if o > 0 else "{}-{}".format(it, -o))


def random_program(seed):
    rng = np.random.default_rng(seed)
    nd = int(rng.integers(1, 4))
    own = ITERATORS[3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(3, 11)), int(rng.integers(3, 13)),
                int(rng.choice([4, 8, 12, 16, 10, 7]))]
    elif nd == 2:
        dims = [int(rng.integers(3, 40)), int(rng.choice([8, 16, 36, 5, 11, 64]))]
    else:
        dims = [int(rng.choice([16, 33, 100]))]
    prog = {"inputs": {}, "outputs": [], "dimensions": dims, "program": {}}
    fields = {}  # name -> (dims list, dtype)
    scalars = []
    n_inputs = int(rng.integers(1, 4))
    for n in range(n_inputs):
        name = "in{}".format(n)
        dtype = str(rng.choice(["float32", "float64"]))
        if n > 0 and nd > 1 and rng.random() < 0.3:
            keep = sorted(rng.choice(nd, size=int(rng.integers(1, nd)), replace=False))
            fdims = [own[d] for d in keep]
        else:
            fdims = list(own)
        prog["inputs"][name] = {"data": "constant:1.0", "data_type": dtype}
        if fdims != list(own):
            prog["inputs"][name]["input_dims"] = fdims
        fields[name] = (fdims, dtype)
    if rng.random() < 0.6:
        prog["inputs"]["s0"] = {"data": float(np.round(rng.uniform(-2, 2), 3)),
                                "data_type": str(rng.choice(["float32", "float64"])),
                                "input_dims": []}
        scalars.append("s0")
    if rng.random() < 0.4:
        prog["constants"] = {"kc": {"value": float(np.round(rng.uniform(-1, 1), 2)),
                                    "data_type": "float64"}}
        scalars.append("kc")

    n_kernels = int(rng.integers(1, 6))
    consumed = set()
    for kn in range(n_kernels):
        kname = "k{}".format(kn)
        candidates = list(fields)
        n_reads = int(rng.integers(1, min(3, len(candidates)) + 1))
        # prefer reading the previous kernel so chains (and fusion) occur
        reads = []
        if kn > 0 and rng.random() < 0.8:
            reads.append("k{}".format(kn - 1))
        while len(reads) < n_reads:
            c = str(rng.choice(candidates))
            if c not in reads:
                reads.append(c)
        bcs, terms = {}, []
        for f in reads:
            fdims, _ = fields[f]
            consumed.add(f)
            kind = rng.random()
            if kind < 0.15:
                bcs[f] = {"type": "shrink"}
            elif kind < 0.45:
                bcs[f] = {"type": "constant", "value": int(rng.integers(-2, 3))}
            else:
                bcs[f] = {"type": "constant", "value": float(np.round(rng.uniform(-1, 1), 2))}
            star = rng.random() < 0.6
            for _ in range(int(rng.integers(1, 5))):
                offs = [0] * len(fdims)
                if star:
                    d = int(rng.integers(0, len(fdims)))
                    offs[d] = int(rng.choice([-1, 0, 1]))
                else:
                    for d in range(len(fdims)):
                        offs[d] = int(rng.choice([-2, -1, 0, 0, 1, 2, 13]))
                terms.append("{}[{}]".format(f, ", ".join(
                    _offset_text(it, o) for it, o in zip(fdims, offs))))
        def leaf():
            r = rng.random()
            if r < 0.7 or not scalars:
                return str(rng.choice(terms))
            if r < 0.85:
                return str(rng.choice(scalars))
            return repr(float(np.round(rng.uniform(-2, 2), 3))) if rng.random() < 0.7 \
                else str(int(rng.integers(1, 4)))
        def expr(depth):
            if depth == 0 or rng.random() < 0.25:
                return leaf()
            r = rng.random()
            if r < 0.6:
                return "({} {} {})".format(expr(depth - 1), rng.choice(["+", "-", "*"]),
                                           expr(depth - 1))
            if r < 0.72:
                return "(-{})".format(expr(depth - 1))
            if r < 0.84:
                return "{}({}, {})".format(rng.choice(["min", "max"]), expr(depth - 1),
                                           expr(depth - 1))
            cond = "{} {} {}".format(expr(depth - 1), rng.choice(["<", "<=", ">", ">=", "=="]),
                                     expr(depth - 1))
            if rng.random() < 0.3:
                cond = "({}) {} ({} > 0.1)".format(cond, rng.choice(["and", "or"]), leaf())
            return "({} if {} else {})".format(expr(depth - 1), cond, expr(depth - 1))
        # make sure every read field is actually used
        body = " + ".join(["0.5 * " + t for t in terms[:1]] + [expr(3)])
        used = [f for f in reads if (f + "[") in body]
        for f in reads:
            if f not in used:
                fdims, _ = fields[f]
                body += " + {}[{}]".format(f, ", ".join(fdims))
        if rng.random() < 0.3:
            text = "tmp = {}; {} = tmp * 0.25 + {}".format(body, kname, leaf())
            for f in reads:  # `leaf` may have named only already-used fields
                pass
        else:
            text = "{} = {}".format(kname, body)
        prog["program"][kname] = {
            "computation_string": text,
            "boundary_conditions": bcs,
            "data_type": str(rng.choice(["float32", "float64"])),
        }
        fields[kname] = (list(own), prog["program"][kname]["data_type"])
    # outputs: the last kernel plus every kernel nobody reads (no orphans)
    outs = []
    for kn in range(n_kernels):
        kname = "k{}".format(kn)
        read_by_later = any((kname + "[") in prog["program"]["k{}".format(m)]["computation_string"]
                            for m in range(kn + 1, n_kernels))
        if kn == n_kernels - 1 or not read_by_later or rng.random() < 0.2:
            outs.append(kname)
    prog["outputs"] = outs
    # drop inputs nobody reads (the reference would leave them dangling)
    text_all = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in list(prog["inputs"]):
        if name.startswith("in") and (name + "[") not in text_all:
            del prog["inputs"][name]
    if "s0" in prog["inputs"] and "s0" not in text_all:
        del prog["inputs"]["s0"]
    if "constants" in prog and "kc" not in text_all:
        del prog["constants"]
    return prog


def random_inputs(prog, seed):
    """Random arrays for the array inputs, the declared value for scalars."""
    from oracle import numpy_oracle as npo
    rng = np.random.default_rng(seed)
    p = npo.load_program(prog)
    vals = {}
    for name, desc in p["inputs"].items():
        dims = npo._input_dims(p, name)
        if dims:
            shape = npo._dims_shape(p, dims)
            vals[name] = rng.uniform(-1, 1, shape).astype(npo._NP[desc["data_type"]])
        else:
            vals[name] = desc["data"]
    return vals


# --- random chains of radius-1 star operators (the fused plane-streaming kernel's
# shape) on awkward domain sizes; only + - * and selects --------------------------
EXACT = [0.0, 0.25, -1.5, 1.0, 0.5, -0.125, 2.0]


def star_program(seed):
    rng = np.random.default_rng(seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(3, 25)), int(rng.integers(3, 41)), 4 * int(rng.integers(2, 41))]
    else:
        dims = [int(rng.integers(3, 120)), 4 * int(rng.integers(2, 80))]
    dtype = "float32" if rng.random() < 0.6 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    if rng.random() < 0.4:
        prog["inputs"]["p"] = {"data": "constant:0.5",
                               "data_type": "float32" if rng.random() < 0.5 else "float64"}
    scalars = []
    for n in range(int(rng.integers(0, 3))):
        name = "s%d" % n
        prog["inputs"][name] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype,
                                "input_dims": []}
        scalars.append(name)
    stages = int(rng.integers(1, 7))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        offs = []
        for d in range(nd):
            for o in (-1, 1):
                if rng.random() < 0.75:
                    offs.append((d, o))
        if not offs:
            offs.append((int(rng.integers(0, nd)), int(rng.choice([-1, 1]))))
        if rng.random() < 0.6:
            offs.append((0, 0))
        order = rng.permutation(len(offs))
        terms = []
        for t in order:
            d, o = offs[int(t)]
            idx = [it if e != d or o == 0 else "%s%+d" % (it, o) for e, it in enumerate(its)]
            acc = "%s[%s]" % (prev, ",".join(idx))
            r = rng.random()
            if r < 0.5:
                terms.append(acc)
            elif r < 0.8 or not scalars:
                terms.append("%r*%s" % (float(np.round(rng.uniform(-1, 1), 4)), acc))
            else:
                terms.append("%s*%s" % (rng.choice(scalars), acc))
        expr = terms[0]
        for t in terms[1:]:
            expr = "%s %s %s" % (expr, rng.choice(["+", "+", "-"]), t)
            if rng.random() < 0.3:
                expr = "(" + expr + ")"
        bcs = {}
        extra = []
        if "p" in prog["inputs"] and rng.random() < 0.4:
            extra.append("p")
        if s >= 2 and rng.random() < 0.2:
            extra.append("b%d" % int(rng.integers(0, s - 1)))
        if prev != "a" and rng.random() < 0.15:
            extra.append("a")
        for f in extra:
            expr = "%s %s %s[%s]" % (expr, rng.choice(["+", "-", "*"]), f, ",".join(its))
            bcs[f] = {"type": "constant", "value": 0.0}
        coef = rng.random()
        if coef < 0.5:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.1, 0.3), 8)), expr)
        elif coef < 0.65 and scalars:
            expr = "%s * (%s)" % (rng.choice(scalars), expr)
        if rng.random() < 0.15:
            centre = "%s[%s]" % (prev, ",".join(its))
            expr = "(%s) if %s > 0.0 else (%s + 1)" % (expr, centre, expr)
        kind = rng.random()
        if kind < 0.1:
            bcs[prev] = {"type": "shrink"}
        elif kind < 0.35:
            bcs[prev] = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bcs[prev] = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": bcs, "data_type": dtype}
        if s < stages - 1 and rng.random() < 0.1:
            prog["outputs"].append(name)
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in list(prog["inputs"]):
        if name != "a" and (name + "[") not in text and name not in scalars:
            del prog["inputs"][name]
    for name in scalars:
        if name not in text:
            del prog["inputs"][name]
    return prog


def dag_program(seed):
    """Fork / join programs of radius-1 star operators (the structure bin/synthesize.py -fork_frequency emits,
    randomised): chains that fork into two branches of random length which a two-field operator joins again,
    intermediates with several readers, intermediate program outputs, an optional auxiliary field read at the
    point itself -- what the DAG groups of kernels/star3d.h fuse (round 4).  Boundary constants of the readers
    of one field mostly agree (the condition for sharing a register window), sometimes not."""
    rng = np.random.default_rng(seed + 4242)
    nd = 3 if rng.random() < 0.55 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(5, 25)), int(rng.integers(3, 41)), 4 * int(rng.integers(2, 41))]
    else:
        dims = [int(rng.integers(6, 120)), 4 * int(rng.integers(2, 80))]
    dtype = "float32" if rng.random() < 0.65 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    use_aux = rng.random() < 0.3
    if use_aux:
        prog["inputs"]["p"] = {"data": "constant:0.5", "data_type": dtype}
    field_bc = {}  # field -> the constant most of its readers declare

    def bc_of(field):
        if field not in field_bc:
            field_bc[field] = int(rng.integers(0, 2)) if rng.random() < 0.3 else float(rng.choice(EXACT))
        if rng.random() < 0.08:  # a reader that disagrees: such operators cannot share the field's window
            return float(rng.choice(EXACT))
        return field_bc[field]

    def star_of(field):
        offs = [(d, o) for d in range(nd) for o in (-1, 1) if rng.random() < 0.7]
        if not offs:
            offs.append((int(rng.integers(0, nd)), int(rng.choice([-1, 1]))))
        if rng.random() < 0.5:
            offs.append((0, 0))
        terms = []
        for t in rng.permutation(len(offs)):
            d, o = offs[int(t)]
            idx = [it if e != d or o == 0 else "%s%+d" % (it, o) for e, it in enumerate(its)]
            acc = "%s[%s]" % (field, ",".join(idx))
            terms.append(acc if rng.random() < 0.6 else "%r*%s" % (float(np.round(rng.uniform(-1, 1), 4)), acc))
        expr = terms[0]
        for t in terms[1:]:
            expr = "%s %s %s" % (expr, rng.choice(["+", "+", "-"]), t)
        return expr

    def add(name, sources):
        expr = " + ".join("(%s)" % star_of(f) for f in sources)
        bcs = {f: {"type": "constant", "value": bc_of(f)} for f in sources}
        if use_aux and rng.random() < 0.3:
            expr = "%s + p[%s]" % (expr, ",".join(its))
            bcs["p"] = {"type": "constant", "value": 0.0}
        expr = "%r * (%s)" % (float(np.round(rng.uniform(0.05, 0.3), 8)), expr)
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr), "boundary_conditions": bcs,
                                 "data_type": dtype}
        if rng.random() < 0.08:
            prog["outputs"].append(name)

    prev, count = "a", 0
    for section in range(int(rng.integers(1, 4))):
        for _ in range(int(rng.integers(0, 3))):  # a piece of chain
            name = "c%d" % count
            count += 1
            add(name, [prev])
            prev = name
        kind = rng.random()
        if kind < 0.7:  # fork into two branches, joined again
            ends = []
            for br in "xy":
                cur = prev
                for d in range(int(rng.integers(1, 3))):
                    name = "f%d%s%d" % (section, br, d)
                    add(name, [cur])
                    cur = name
                ends.append(cur)
            name = "j%d" % section
            if rng.random() < 0.85:
                add(name, ends)
            else:  # no join: both branch ends are results
                prog["outputs"].extend(e for e in ends if e not in prog["outputs"])
                add(name, [ends[0]])
            prev = name
        elif kind < 0.85:  # an intermediate with two readers that are both results
            name = "m%d" % section
            add(name, [prev])
            for br in "xy":
                add("r%d%s" % (section, br), [name])
                prog["outputs"].append("r%d%s" % (section, br))
            prev = name
    if prev == "a":
        add("c%d" % count, ["a"])
        prev = "c%d" % count
    if prev not in prog["outputs"]:
        prog["outputs"].append(prev)
    prog["outputs"] = list(dict.fromkeys(prog["outputs"]))
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    if use_aux and "p[" not in text:
        del prog["inputs"]["p"]
    return prog


def with_copy_boundaries(prog, seed, share=0.6):
    """The program with the boundary condition of a random share of its operators' streamed
    fields (those read off-centre) turned into `copy` -- an out-of-domain read takes the value at
    the point itself (reference stencil/intel_fpga.py:225-227).  A generator of its own, so that
    the programs of the other generators keep their seeds.  (The reference's CPU expansion
    raises for `copy`, stencil/cpu.py:87, and so do the oracles: these programs are checked
    HIP against HIP -- fused against the one-operator-per-launch generic kernel -- and against a
    NumPy statement of the rule for the Jacobi chain, tests/test_gpu_parity.py.)"""
    import copy as _copy
    rng = np.random.default_rng([seed, 77])
    out = _copy.deepcopy(prog)
    changed = 0
    for name, k in out["program"].items():
        for f, bc in k["boundary_conditions"].items():
            if (f == "a" or f.startswith("b")) and rng.random() < share:
                k["boundary_conditions"][f] = {"type": "copy"}
                changed += 1
    if not changed:
        first = next(iter(out["program"].values()))
        f = next(iter(first["boundary_conditions"]))
        first["boundary_conditions"][f] = {"type": "copy"}
    return out


# --- random chains of WIDE-STAR operators (kernels/wstar3d.h): offsets -2..2 along one axis
# at a time (at least one at distance 2 in most stages; a stage without one is a plain star and
# breaks the group), scalar / literal coefficients, int / float / shrink boundaries, ternaries
# on the centre value; 3-D and 2-D, awkward sizes (rows of 4m and 4m+2 elements); only + - *
# and selects, so every implementation must agree bit for bit -------------------------------
def wide_program(seed):
    rng = np.random.default_rng(10_000 + seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(5, 30)), int(rng.integers(3, 45)), 2 * int(rng.integers(4, 90))]
    else:
        dims = [int(rng.integers(5, 140)), 2 * int(rng.integers(4, 200))]
    dtype = "float32" if rng.random() < 0.65 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    scalars = []
    for n in range(int(rng.integers(0, 3))):
        name = "s%d" % n
        prog["inputs"][name] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype, "input_dims": []}
        scalars.append(name)
    stages = int(rng.integers(1, 6))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        reach = 2 if rng.random() < 0.85 else 1
        offs = []
        for d in range(nd):
            for o in range(-reach, reach + 1):
                if o != 0 and rng.random() < 0.7:
                    offs.append((d, o))
        if reach == 2 and not any(abs(o) == 2 for _, o in offs):
            offs.append((int(rng.integers(0, nd)), int(rng.choice([-2, 2]))))
        if not offs:
            offs.append((int(rng.integers(0, nd)), 1))
        if rng.random() < 0.6:
            offs.append((0, 0))
        terms = []
        for t in rng.permutation(len(offs)):
            d, o = offs[int(t)]
            idx = [it if e != d or o == 0 else "%s%+d" % (it, o) for e, it in enumerate(its)]
            acc = "%s[%s]" % (prev, ",".join(idx))
            r = rng.random()
            if r < 0.5:
                terms.append(acc)
            elif r < 0.8 or not scalars:
                terms.append("%r*%s" % (float(np.round(rng.uniform(-1, 1), 4)), acc))
            else:
                terms.append("%s*%s" % (rng.choice(scalars), acc))
        expr = terms[0]
        for t in terms[1:]:
            expr = "%s %s %s" % (expr, rng.choice(["+", "+", "-"]), t)
            if rng.random() < 0.25:
                expr = "(" + expr + ")"
        coef = rng.random()
        if coef < 0.5:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.05, 0.2), 8)), expr)
        elif coef < 0.65 and scalars:
            expr = "%s * (%s)" % (rng.choice(scalars), expr)
        if rng.random() < 0.15:
            centre = "%s[%s]" % (prev, ",".join(its))
            expr = "(%s) if %s > 0.0 else (%s + 1)" % (expr, centre, expr)
        kind = rng.random()
        if kind < 0.1:
            bc = {"type": "shrink"}
        elif kind < 0.4:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        if s < stages - 1 and rng.random() < 0.1:
            prog["outputs"].append(name)
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in scalars:
        if name not in text:
            del prog["inputs"][name]
    return prog


# --- random chains of DENSE-neighbourhood operators (kernels/dense3d.h): any subset of the
# offsets {-2..2}^3 (2-D: {-2..2}^2) of the previous stage, at least one of them beyond what the
# star / compact / wide-star kernels take (a diagonal at distance 2), scalar / literal
# coefficients, int / float / shrink boundaries; rows of 4m elements; only + - * and selects ------
def dense_program(seed):
    rng = np.random.default_rng(20_000 + seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(5, 22)), int(rng.integers(3, 50)), 4 * int(rng.integers(2, 50))]
    else:
        dims = [int(rng.integers(5, 120)), 4 * int(rng.integers(2, 150))]
    dtype = "float32" if rng.random() < 0.65 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    scalars = []
    for n in range(int(rng.integers(0, 3))):
        name = "s%d" % n
        prog["inputs"][name] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype, "input_dims": []}
        scalars.append(name)
    stages = int(rng.integers(1, 4))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        density = rng.choice([0.08, 0.3, 1.0])
        offs = []
        for idx in np.ndindex(*([5] * nd)):
            off = tuple(int(x) - 2 for x in idx)
            if any(off) and rng.random() < density:
                offs.append(off)
        far = tuple(int(rng.choice([-2, 2])) if d < 2 else int(rng.choice([-2, -1, 1, 2])) for d in range(nd))
        if far not in offs:
            offs.append(far)  # a diagonal at distance 2: beyond every other fused kernel
        if rng.random() < 0.6:
            offs.append((0, ) * nd)
        terms = []
        for t in rng.permutation(len(offs)):
            off = offs[int(t)]
            idx = [it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off)]
            acc = "%s[%s]" % (prev, ",".join(idx))
            r = rng.random()
            if r < 0.6:
                terms.append(acc)
            elif r < 0.85 or not scalars:
                terms.append("%r*%s" % (float(np.round(rng.uniform(-1, 1), 4)), acc))
            else:
                terms.append("%s*%s" % (rng.choice(scalars), acc))
        expr = terms[0]
        for t in terms[1:]:
            expr = "%s %s %s" % (expr, rng.choice(["+", "+", "-"]), t)
            if rng.random() < 0.1:
                expr = "(" + expr + ")"
        if rng.random() < 0.6:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.01, 0.1), 8)), expr)
        kind = rng.random()
        if kind < 0.1:
            bc = {"type": "shrink"}
        elif kind < 0.4:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in scalars:
        if name not in text:
            del prog["inputs"][name]
    return prog


def dense_sum_program(seed):
    """Operators that are ONE left-associated sum over a random subset of {-2..2}^d, the terms in
    RANDOM order (the generator's boxes list them in lexicographic order), optionally scaled once by a
    literal or a scalar; the centre (a float operand among doubles when the boundary literal is a
    float) anywhere in the sum, also first: the plain-sum form of the dense kernel (codegen.hpp:
    dense_sum_form) or, where the partial sums would change type, the text as it stands."""
    rng = np.random.default_rng(30_000 + seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(5, 22)), int(rng.integers(3, 50)), 4 * int(rng.integers(2, 50))]
    else:
        dims = [int(rng.integers(5, 120)), 4 * int(rng.integers(2, 150))]
    dtype = "float32" if rng.random() < 0.65 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    if rng.random() < 0.4:
        prog["inputs"]["s0"] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype, "input_dims": []}
    stages = int(rng.integers(1, 4))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        density = rng.choice([0.1, 0.4, 1.0])
        offs = [tuple(int(x) - 2 for x in idx) for idx in np.ndindex(*([5] * nd))
                if any(int(x) != 2 for x in idx) and rng.random() < density]
        far = tuple(int(rng.choice([-2, 2])) if d < 2 else int(rng.choice([-2, -1, 1, 2])) for d in range(nd))
        # (round 4; its own generator again) one operator in four reaches THREE points: subsets of {-3..3}^d -- ordered
        # by plane the dense kernel's streaming form with seven open planes, otherwise the generic kernel
        r3 = np.random.default_rng(78_000 + 11 * seed + s)
        if r3.random() < 0.25:
            dens3 = float(r3.choice([0.03, 0.15, 0.5]))
            offs = [tuple(int(x) - 3 for x in idx) for idx in np.ndindex(*([7] * nd))
                    if any(int(x) != 3 for x in idx) and r3.random() < dens3]
            far = tuple(int(r3.choice([-3, 3])) for _ in range(nd))
        if far not in offs:
            offs.append(far)
        if rng.random() < 0.7:
            offs.append((0, ) * nd)
        if rng.random() < 0.5:
            offs = [offs[int(t)] for t in rng.permutation(len(offs))]
        # (round 4; its own generator, so that the programs of earlier campaigns keep their seeds) two operators in
        # five have their terms ordered by plane -- any order inside a plane --: the dense kernel's streaming form
        if np.random.default_rng(77_000 + 7 * seed + s).random() < 0.4:
            offs.sort(key=lambda o: o[0])
        terms = ["%s[%s]" % (prev, ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off))) for off in offs]
        expr = " + ".join(terms)
        c = rng.random()
        if c < 0.5:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.01, 0.1), 8)), expr)
        elif c < 0.7 and "s0" in prog["inputs"]:
            expr = "s0 * (%s)" % expr
        kind = rng.random()
        if kind < 0.1:
            bc = {"type": "shrink"}
        elif kind < 0.4:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    if "s0" in prog["inputs"] and "s0 *" not in text:
        del prog["inputs"]["s0"]
    return prog


def box_sum_program(seed):
    """Chains of 2-6 operators, each ONE left-associated sum over a random subset of {-1,0,1}^d ordered by plane (any
    order inside a plane), optionally scaled once; constant boundaries (int or float literal), now and then an
    operator with `shrink` or with its terms in random order in between (those end a fused pair): what the dense
    kernel's fused streaming form takes two at a time (round 4: dense3d.h SF_DENSE_T2, plan option dense.t2)."""
    rng = np.random.default_rng(91_000 + seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(4, 26)), int(rng.integers(3, 60)), 4 * int(rng.integers(2, 140))]
    else:
        dims = [int(rng.integers(5, 140)), 4 * int(rng.integers(2, 300))]
    dtype = "float32" if rng.random() < 0.7 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    if rng.random() < 0.4:
        prog["inputs"]["s0"] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype, "input_dims": []}
    stages = int(rng.integers(2, 7))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        density = rng.choice([0.3, 0.7, 1.0])
        offs = [tuple(int(x) - 1 for x in idx) for idx in np.ndindex(*([3] * nd))
                if any(int(x) != 1 for x in idx) and rng.random() < density]
        if not offs:
            offs = [tuple([1] + [0] * (nd - 1))]
        if rng.random() < 0.8:
            offs.append((0, ) * nd)
        offs = [offs[int(t)] for t in rng.permutation(len(offs))]
        if rng.random() < 0.85:
            offs.sort(key=lambda o: o[0])
        terms = ["%s[%s]" % (prev, ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off))) for off in offs]
        if len(terms) < 2:
            terms.append("%s[%s]" % (prev, ",".join(its)))
        expr = " + ".join(terms)
        c = rng.random()
        if c < 0.5:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.01, 0.3), 8)), expr)
        elif c < 0.7 and "s0" in prog["inputs"]:
            expr = "s0 * (%s)" % expr
        kind = rng.random()
        if kind < 0.08:
            bc = {"type": "shrink"}
        elif kind < 0.5:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        prev = name
    prog["outputs"].append(prev)
    if rng.random() < 0.3 and stages >= 3:
        prog["outputs"].append("b%d" % int(rng.integers(0, stages - 1)))  # an intermediate that is also an output
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    if "s0" in prog["inputs"] and "s0 *" not in text:
        del prog["inputs"]["s0"]
    return prog


def sparse_sum_program(seed):
    """Chains of 2-6 operators, each ONE left-associated sum of at most 16 terms within two points of the centre: the
    generator's radius-2 cross in its own order (i-2 .. i+2 first, then the in-plane terms -- which meet their output
    plane two steps after their own plane arrived), the same cross shuffled, or a random sparse subset of {-2..2}^d in
    random order; optionally scaled once; constant boundaries, now and then `shrink` or a 27-point operator in between
    (those end a fused pair).  What the dense kernel's fused streaming form takes two at a time since round 5 (dense3d.h:
    SF_RS 2, SF_LAG / SF_LAG2; planner: select_dense_t2)."""
    rng = np.random.default_rng(123_000 + seed)
    nd = 3 if rng.random() < 0.85 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(4, 26)), int(rng.integers(3, 60)), 4 * int(rng.integers(2, 140))]
    else:
        dims = [int(rng.integers(5, 140)), 4 * int(rng.integers(2, 300))]
    dtype = "float32" if rng.random() < 0.8 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    if rng.random() < 0.4:
        prog["inputs"]["s0"] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype, "input_dims": []}
    stages = int(rng.integers(2, 7))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        shape = rng.random()
        cross = [tuple(d if a == ax else 0 for a in range(nd)) for ax in range(nd) for d in (-2, -1, 1, 2)]
        if shape < 0.3:
            offs = cross
        elif shape < 0.5:
            offs = [cross[int(t)] for t in rng.permutation(len(cross))]
        elif shape < 0.92:
            count = int(rng.integers(2, 16))
            cells = [tuple(int(x) - 2 for x in idx) for idx in np.ndindex(*([5] * nd)) if any(int(x) != 2 for x in idx)]
            offs = [cells[int(t)] for t in rng.permutation(len(cells))[:count]]
            if not any(2 in map(abs, o) for o in offs):
                offs.append(tuple(int(rng.choice([-2, 2])) if d == 0 else 0 for d in range(nd)))
        else:  # a 27-point (9-point) sum ordered by plane: not one of the sparse ones
            offs = [tuple(int(x) - 1 for x in idx) for idx in np.ndindex(*([3] * nd))]
        if shape < 0.92 and rng.random() < 0.5:
            offs.insert(int(rng.integers(0, len(offs) + 1)), (0, ) * nd)
        offs = list(dict.fromkeys(offs))
        terms = ["%s[%s]" % (prev, ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off))) for off in offs]
        expr = " + ".join(terms)
        c = rng.random()
        if c < 0.5:
            expr = "%r * (%s)" % (float(np.round(rng.uniform(0.01, 0.3), 8)), expr)
        elif c < 0.7 and "s0" in prog["inputs"]:
            expr = "s0 * (%s)" % expr
        kind = rng.random()
        if kind < 0.06:
            bc = {"type": "shrink"}
        elif kind < 0.5:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        prev = name
    prog["outputs"].append(prev)
    if rng.random() < 0.3 and stages >= 3:
        prog["outputs"].append("b%d" % int(rng.integers(0, stages - 1)))  # an intermediate that is also an output
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    if "s0" in prog["inputs"] and "s0 *" not in text:
        del prog["inputs"]["s0"]
    return prog


def weighted_cross_program(seed):
    """Chains of 2-6 star operators, each ONE left-associated sum of products `factor * a[...]` / `a[...] * factor` / plain
    accesses over a cross of radius 1 or 2 (the generator's `diffusion` shapes: a scalar per term, the centre first -- or
    anywhere, or absent), factors literals or scalars of the program, the terms in the generator's order or shuffled;
    constant boundaries.  What the dense kernel's fused streaming forms take with a factor per term since round 5 (three
    per launch at radius 1, two at radius 2; dense.t2=3 forces the forms onto small grids)."""
    rng = np.random.default_rng(456_000 + seed)
    nd = 3 if rng.random() < 0.9 else 2
    its = ["i", "j", "k"][3 - nd:]
    if nd == 3:
        dims = [int(rng.integers(4, 26)), int(rng.integers(3, 60)), 4 * int(rng.integers(2, 140))]
    else:
        dims = [int(rng.integers(5, 140)), 4 * int(rng.integers(2, 300))]
    dtype = "float32" if rng.random() < 0.85 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    for name in ("s0", "s1"):
        prog["inputs"][name] = {"data": float(np.round(rng.uniform(-0.5, 0.5), 3)), "data_type": dtype, "input_dims": []}
    stages = int(rng.integers(2, 7))
    radius = 1 if rng.random() < 0.5 else 2
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        if rng.random() < 0.15:
            radius = 3 - radius  # (a change of radius ends a fused group)
        offs = [tuple(d if a == ax else 0 for a in range(nd)) for ax in range(nd) for d in range(-radius, radius + 1) if d]
        if rng.random() < 0.4:
            offs = [offs[int(t)] for t in rng.permutation(len(offs))]
        if rng.random() < 0.7:
            offs.insert(0 if rng.random() < 0.6 else int(rng.integers(0, len(offs) + 1)), (0, ) * nd)
        terms = []
        weights = rng.random() < 0.85  # (now and then an operator without factors: a plain sum in the chain)
        for off in offs:
            acc = "%s[%s]" % (prev, ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off)))
            kind = rng.random()
            if not weights or kind < 0.15:
                terms.append(acc)
            else:
                factor = ("s0" if kind < 0.45 else "s1" if kind < 0.6 else repr(float(np.round(rng.uniform(0.05, 0.4), 6))))
                terms.append("%s * %s" % (factor, acc) if rng.random() < 0.8 else "%s * %s" % (acc, factor))
        expr = " + ".join(terms)
        kind = rng.random()
        if kind < 0.5:
            bc = {"type": "constant", "value": int(rng.integers(-1, 3))}
        else:
            bc = {"type": "constant", "value": float(rng.choice(EXACT))}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": {prev: bc}, "data_type": dtype}
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in ("s0", "s1"):
        if name + " *" not in text and "* " + name not in text:
            del prog["inputs"][name]
    return prog


# --- random chains of COMPACT operators (kernels/compact3d.h): any subset of the 27
# offsets {-1,0,1}^3 of the previous stage, optionally a second full input field read
# through such offsets, scalar / literal coefficients, int / float / shrink boundaries,
# plain star stages in between; 3-D and 2-D, awkward sizes; only + - * and selects -----
def compact_program(seed):
    rng = np.random.default_rng(seed)
    nd = 3 if rng.random() < 0.7 else 2
    its = ["i", "j", "k"][3 - nd:]
    vk = int(rng.choice([4, 4, 4, 2, 1]))
    if nd == 3:
        dims = [int(rng.integers(3, 22)), int(rng.integers(3, 37)), vk * int(rng.integers(2, 36))]
    else:
        dims = [int(rng.integers(3, 90)), vk * int(rng.integers(2, 150))]
    if vk == 1 and dims[-1] % 2 == 0:
        dims[-1] += 1
    dtype = "float32" if rng.random() < 0.65 else "float64"
    prog = {"inputs": {"a": {"data": "constant:1.0", "data_type": dtype}}, "outputs": [],
            "dimensions": dims, "program": {}}
    extras = []
    for n in range(int(rng.integers(0, 4))):
        name = "e%d" % n
        prog["inputs"][name] = {"data": "constant:0.5", "data_type": dtype}
        extras.append(name)
    scalars = []
    for n in range(int(rng.integers(0, 3))):
        name = "s%d" % n
        prog["inputs"][name] = {"data": float(np.round(rng.uniform(-1, 1), 3)), "data_type": dtype,
                                "input_dims": []}
        scalars.append(name)

    def access(field, off):
        return "%s[%s]" % (field, ",".join(it if o == 0 else "%s%+d" % (it, o) for it, o in zip(its, off)))

    def offsets(density, allow_diagonal=True):
        import itertools
        offs = []
        for off in itertools.product((-1, 0, 1), repeat=nd):
            nz = sum(1 for o in off if o != 0)
            if nz > 1 and not allow_diagonal:
                continue
            if rng.random() < density:
                offs.append(tuple(off))
        if not offs:
            offs.append(tuple(int(v) for v in rng.integers(-1, 2, nd)))
        return [offs[int(t)] for t in rng.permutation(len(offs))]

    def bc():
        kind = rng.random()
        if kind < 0.1:
            return {"type": "shrink"}
        if kind < 0.35:
            return {"type": "constant", "value": int(rng.integers(-1, 3))}
        return {"type": "constant", "value": float(rng.choice(EXACT))}

    stages = int(rng.integers(1, 7))
    prev = "a"
    for s in range(stages):
        name = "b%d" % s
        style = rng.random()
        density = 1.0 if style < 0.2 else float(rng.uniform(0.15, 0.7))
        fields = [(prev, offsets(density, allow_diagonal=style < 0.85))]
        if extras and rng.random() < 0.5:
            fields.append((str(rng.choice(extras)), offsets(float(rng.uniform(0.1, 0.5)), allow_diagonal=rng.random() < 0.6)))
        terms = []
        for f, offs in fields:
            for off in offs:
                r = rng.random()
                if r < 0.6:
                    terms.append(access(f, off))
                elif r < 0.85 or not scalars:
                    terms.append("%r*%s" % (float(np.round(rng.uniform(-1, 1), 4)), access(f, off)))
                else:
                    terms.append("%s*%s" % (rng.choice(scalars), access(f, off)))
        terms = [terms[int(t)] for t in rng.permutation(len(terms))]
        expr = terms[0]
        for t in terms[1:]:
            expr = "%s %s %s" % (expr, rng.choice(["+", "+", "-"]), t)
            if rng.random() < 0.2:
                expr = "(" + expr + ")"
        if rng.random() < 0.6:
            expr = "%r * (%s)" % (float(np.round(1.0 / max(1, len(terms)), 8)), expr)
        if rng.random() < 0.1:
            expr = "(%s) if %s > 0.0 else (%s - 0.5)" % (expr, access(prev, (0, ) * nd), expr)
        bcs = {f: bc() for f, _ in fields}
        prog["program"][name] = {"computation_string": "%s = %s" % (name, expr),
                                 "boundary_conditions": bcs, "data_type": dtype}
        if s < stages - 1 and rng.random() < 0.1:
            prog["outputs"].append(name)
        prev = name
    prog["outputs"].append(prev)
    text = " ".join(k["computation_string"] for k in prog["program"].values())
    for name in extras:
        if (name + "[") not in text:
            del prog["inputs"][name]
    for name in scalars:
        if name not in text:
            del prog["inputs"][name]
    return prog
