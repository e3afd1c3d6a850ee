"""TEST INFRASTRUCTURE — C/OpenMP restatement of StencilFlow's CPU reference.

Never imported by the product package; used by ``tests/`` (cross-checked
bit-for-bit against ``numpy_oracle``) and by the ``cpu_baseline`` leg of
``bench.py`` (``kind: "port"``).

The reference's CPU program is C++ that DaCe generates from the program
description and compiles at run time (stencilflow/run_program.py:92,127-128;
stencilflow/sdfg_generator.py:580-677; stencilflow/stencil/cpu.py:19-191).
DaCe is not available, so this module emits the equivalent
plain C itself — one full-domain loop nest per operator, operators in
topological order (sdfg_generator.py:638-675), per point

    <access> = <bc> if <out of domain> else <field>[p + offset]   cpu.py:71-102
    <the kernel's statements>                                      cpu.py:46-52,115
    out[p] = <kernel name>                                         sdfg_generator.py:104-106

and compiles it with ``gcc -O3 -fopenmp -ffp-contract=off`` (strict IEEE; the
reference's own build uses DaCe's default ``-ffast-math``, so its last bits are
compiler-dependent — see numpy_oracle's header).  Types are left to the C
compiler (``__auto_type`` locals, ``<tgmath.h>`` calls), which makes this an
independent check of the explicit typing in ``numpy_oracle``.

PARITY: pinned through numpy_oracle.py (bit-identical to it on every fixture and
on random programs), whose header lists the vectors captured from the reference.
"""

import ast
import ctypes
import hashlib
import os
import subprocess

import numpy as np

from . import numpy_oracle as npo

ITERATORS = npo.ITERATORS
_CT = {"float32": "float", "float64": "double", "int32": "int",
       "int64": "long long"}
_BUILD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build")

_PRELUDE = r"""
#include <math.h>
#include <tgmath.h>
#include <stddef.h>
#define SF_MIN(a, b) ((a) < (b) ? (a) : (b))
#define SF_MAX(a, b) ((a) > (b) ? (a) : (b))
/* Python "/" is true division */
#define SF_DIV(a, b) (_Generic((a) + (b), int: (double)(a) / (double)(b), \
    long: (double)(a) / (double)(b), long long: (double)(a) / (double)(b), \
    default: (a) / (b)))
"""


def _lit(v):
    if isinstance(v, bool):
        return "1" if v else "0"
    if isinstance(v, int):
        return str(v) if -2**31 <= v < 2**31 else str(v) + "LL"
    t = repr(float(v))
    if "." not in t and "e" not in t and "n" not in t:
        t += ".0"
    return t


def _var(field, offs):
    # naming of stencilflow/stencil/subscript_converter.py:12-29
    return field + "_" + "_".join(("m" + str(-o)) if o < 0 else str(o)
                                  for o in offs)


class _Emit(ast.NodeVisitor):
    def __init__(self, field_dims, scalar_names):
        self.field_dims = field_dims
        self.scalar_names = scalar_names
        self.accesses = {}  # var -> (field, offsets)

    def visit_Constant(self, n):
        return _lit(n.value)

    def visit_Name(self, n):
        return n.id

    def visit_Subscript(self, n):
        field = n.value.id
        sl = n.slice
        elts = list(sl.elts) if isinstance(sl, ast.Tuple) else [sl]
        offs, order = [], []
        for e in elts:
            if isinstance(e, ast.Name):
                order.append(e.id)
                offs.append(0)
            else:
                c = int(e.right.value)
                order.append(e.left.id)
                offs.append(-c if isinstance(e.op, ast.Sub) else c)
        if order != list(self.field_dims[field]):
            raise ValueError("access does not match field dims: " +
                             ast.unparse(n))
        var = _var(field, offs)
        self.accesses[var] = (field, tuple(offs))
        return var

    def visit_BinOp(self, n):
        a, b = self.visit(n.left), self.visit(n.right)
        if isinstance(n.op, ast.Div):
            return "SF_DIV({}, {})".format(a, b)
        op = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*"}[type(n.op)]
        return "({} {} {})".format(a, op, b)

    def visit_UnaryOp(self, n):
        v = self.visit(n.operand)
        return v if isinstance(n.op, ast.UAdd) else "(-{})".format(v)

    def visit_Compare(self, n):
        op = {ast.Lt: "<", ast.LtE: "<=", ast.Gt: ">", ast.GtE: ">=",
              ast.Eq: "==", ast.NotEq: "!="}[type(n.ops[0])]
        return "({} {} {})".format(self.visit(n.left), op,
                                   self.visit(n.comparators[0]))

    def visit_BoolOp(self, n):
        op = "&&" if isinstance(n.op, ast.And) else "||"
        return "({} {} {})".format(self.visit(n.values[0]), op,
                                   self.visit(n.values[1]))

    def visit_IfExp(self, n):
        return "({} ? {} : {})".format(self.visit(n.test), self.visit(n.body),
                                       self.visit(n.orelse))

    def visit_Call(self, n):
        fn = n.func.id
        args = [self.visit(a) for a in n.args]
        if fn == "min":
            return "SF_MIN({}, {})".format(*args)
        if fn == "max":
            return "SF_MAX({}, {})".format(*args)
        if fn in ("abs", "fabs"):
            return "fabs({})".format(args[0])
        return "{}({})".format(fn, args[0])

    def generic_visit(self, n):
        raise TypeError("unsupported syntax: " + type(n).__name__)


def generate_c(prog):
    """C source with one ``sf_ref_<kernel>`` function per operator."""
    own = npo._own_iterators(prog)
    shape = list(prog["dimensions"])
    field_dims, field_types, scalars = {}, {}, {}
    for name, desc in prog["inputs"].items():
        dims = npo._input_dims(prog, name)
        if dims:
            field_dims[name] = dims
            field_types[name] = desc["data_type"]
        else:
            scalars[name] = desc["data_type"]
    consts = prog.get("constants", {})
    for kname, k in prog["program"].items():
        field_dims[kname] = own
        field_types[kname] = k["data_type"]
    src = [_PRELUDE]
    signatures = {}
    for kname in npo.topological_kernels(prog):
        k = prog["program"][kname]
        em = _Emit(field_dims, scalars)
        tree = ast.parse(k["computation_string"])
        stmts = [(s.targets[0].id, em.visit(s.value)) for s in tree.body]
        reads = []
        for var, (field, offs) in em.accesses.items():
            if field not in reads:
                reads.append(field)
        used_scalars = [s for s in scalars
                        if any(isinstance(n, ast.Name) and n.id == s
                               for n in ast.walk(tree))]
        params = ["const {}* restrict {}_in".format(_CT[field_types[f]], f)
                  for f in reads]
        params.append("{}* restrict {}_out".format(_CT[k["data_type"]], kname))
        params += ["const {} {}".format(_CT[scalars[s]], s)
                   for s in used_scalars]
        signatures[kname] = (reads, used_scalars)
        body = []
        for name, desc in consts.items():
            body.append("const {} {} = {};".format(_CT[desc["data_type"]], name,
                                                   _lit(desc["value"])))
        for var, (field, offs) in em.accesses.items():
            dims = field_dims[field]
            idx = ""
            for d, o in zip(dims, offs):
                ext = shape[own.index(d)]
                term = "({} + ({}))".format(d, o)
                idx = term if not idx else "({}) * {} + {}".format(idx, ext,
                                                                  term)
            conds = []
            for d, o in zip(dims, offs):
                ext = shape[own.index(d)]
                if o < 0:
                    conds.append("{} < {}".format(d, -o))
                elif o > 0:
                    conds.append("{} >= {}".format(d, ext - o))
            load = "{}_in[{}]".format(field, idx)
            if not conds:  # cpu.py:82-84
                body.append("const __auto_type {} = {};".format(var, load))
                continue
            bc = k["boundary_conditions"][field]
            kind = bc.get("type", bc.get("btype"))
            if kind == "constant":
                fill = _lit(bc["value"])
            elif kind == "shrink":
                fill = _lit(npo.JUNK_VAL)
            else:
                raise ValueError(
                    "Unsupported boundary condition type: {}".format(kind))
            body.append("const __auto_type {} = ({}) ? {} : {};".format(
                var, " || ".join(conds), fill, load))
        for target, text in stmts:
            body.append("const __auto_type {} = {};".format(target, text))
        oidx = ""
        for d in own:
            ext = shape[own.index(d)]
            oidx = d if not oidx else "({}) * {} + {}".format(oidx, ext, d)
        body.append("{}_out[{}] = ({}){};".format(kname, oidx,
                                                  _CT[k["data_type"]], kname))
        loops_open, loops_close = "", ""
        for depth, d in enumerate(own):
            loops_open += "for (long {d} = 0; {d} < {n}; ++{d}) ".format(
                d=d, n=shape[depth])
        src.append("void sf_ref_{}({}) {{\n#pragma omp parallel for schedule(static)\n{}{{\n  {}\n}}\n}}\n"
                   .format(kname, ", ".join(params), loops_open,
                           "\n  ".join(body)))
    return "\n".join(src), signatures, field_types, scalars


class CompiledReference:
    """gcc-compiled reference program; ``run`` mirrors ``numpy_oracle.run_reference``."""

    def __init__(self, program, threads=None, opt="-O3"):
        self.prog = npo.load_program(program)
        source, self.signatures, self.field_types, self.scalars = generate_c(
            self.prog)
        os.makedirs(_BUILD_DIR, exist_ok=True)
        tag = hashlib.sha1((source + opt).encode()).hexdigest()[:16]
        self.so_path = os.path.join(_BUILD_DIR, "ref_{}.so".format(tag))
        if not os.path.exists(self.so_path):
            c_path = os.path.join(_BUILD_DIR, "ref_{}.c".format(tag))
            with open(c_path, "w") as f:
                f.write(source)
            cmd = ["gcc", opt, "-march=native", "-fopenmp", "-ffp-contract=off",
                   "-fno-fast-math", "-std=gnu11", "-shared", "-fPIC", c_path,
                   "-o", self.so_path + ".tmp", "-lm"]
            subprocess.check_call(cmd)
            os.replace(self.so_path + ".tmp", self.so_path)
        self.lib = ctypes.CDLL(self.so_path)
        self.threads = threads
        self.source = source

    def run(self, inputs=None, input_directory=None, return_all=False,
            generate_input=False, stages=None):
        prog = self.prog
        inputs = inputs or {}
        fields, scalar_vals = {}, {}
        for name, desc in prog["inputs"].items():
            override = inputs.get(name)
            if override is None and generate_input:
                override = "constant:0.5"
            val = npo.materialise_input(prog, name, input_directory, override)
            if npo._input_dims(prog, name):
                fields[name] = np.ascontiguousarray(val)
            else:
                scalar_vals[name] = val
        order = npo.topological_kernels(prog)
        if stages is not None:
            order = order[:stages]
        last_use = {}
        for i, kname in enumerate(order):
            for f in self.signatures[kname][0]:
                last_use[f] = i
        keep = set(prog["outputs"]) | set(prog["inputs"])
        if return_all:
            keep |= set(order)
        results = {}
        pool = []
        if self.threads:
            # the OpenMP runtime is already initialised: set the team size directly
            self.lib.omp_set_num_threads(int(self.threads))
        for i, kname in enumerate(order):
            reads, used = self.signatures[kname]
            dt = npo._NP[prog["program"][kname]["data_type"]]
            out = None
            for j, cand in enumerate(pool):
                if cand.dtype == dt:
                    out = pool.pop(j)
                    break
            if out is None:
                out = np.empty(tuple(prog["dimensions"]), dtype=dt)
            fn = getattr(self.lib, "sf_ref_" + kname)
            args = [fields[f].ctypes.data_as(ctypes.c_void_p) for f in reads]
            args.append(out.ctypes.data_as(ctypes.c_void_p))
            for s in used:
                ct = ctypes.c_float if self.scalars[s] == "float32" \
                    else ctypes.c_double
                args.append(ct(float(scalar_vals[s])))
            fn.restype = None
            fn(*args)
            fields[kname] = out
            results[kname] = out
            for f in reads:
                if last_use.get(f) == i and f not in keep:
                    pool.append(fields.pop(f))
        if return_all or stages is not None:
            return results if return_all else {order[-1]: results[order[-1]]}
        return {name: results[name] for name in prog["outputs"]}


def run_reference(program, inputs=None, input_directory=None, **kw):
    return CompiledReference(program).run(inputs, input_directory, **kw)
