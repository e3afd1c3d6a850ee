"""TEST INFRASTRUCTURE — CPU restatement of StencilFlow's reference program.

This module is the *oracle* the HIP backend is checked against.  It is never
imported by the product package (``stencilflow_amd``); only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

It restates, with NumPy whole-array operations, what the reference builds with
``generate_reference`` and runs on the CPU when ``-compare-to-reference`` is
given.  Every step cites the reference lines it follows (paths relative to
``/root/reference``):

* program file, dimension handling ........ stencilflow/kernel_chain_graph.py:364-407
* input materialisation ................... stencilflow/helper.py:162-237
* which arrays exist, execution order ..... stencilflow/sdfg_generator.py:580-677
* accesses, BC records per operator ....... stencilflow/sdfg_generator.py:68-176
* per-point semantics (the arithmetic) .... stencilflow/stencil/cpu.py:58-115,141-169
* ``shrink`` junk value ................... stencilflow/stencil/_common.py:8
* comparison rule ......................... stencilflow/helper.py:261-276

PARITY PINNING.  The reference's CPU path cannot be built in this environment:
it is C++ generated at run time by DaCe (un-vendored, empty submodule ``dace/``,
no version recoverable), and the reference stores no expected
outputs.  This oracle is therefore pinned by (a) the closed-form known answers
the reference's own test programs imply (tests/test_oracle_kat.py), (b) output
vectors of the reference's own ``Simulator`` -- on simulator12.json, on nine
float32 / mixed-dtype programs and on BASELINE configs[0] itself (jacobi3d 32^3,
8 operators), captured by tests/golden/make_reference_fixtures.py and
make_simulator_fixtures.py -- which this module reproduces BIT FOR BIT when it
types literals the way the Simulator does (``typing="nep50"``,
tests/test_reference_vectors.py), and (c) structure fixtures captured from the
reference's ``KernelChainGraph``.  What no vector from the reference covers is
the TYPING CONTRACT below (the Simulator is NumPy, not DaCe's C++) and iterates
deeper than C1's eight operators: for those, parity is pinned up to that typing
only; tests/rounding_envelope.py measures how far the admissible typings lie
apart.

TYPING CONTRACT.  The reference CPU kernel is the C++ DaCe emits for the Python
tasklet text; Python ``float`` literals print as C++ ``double`` literals, ``int``
literals as ``int``, float32 loads are ``float`` and a boundary select
``bc if oob else x`` is a C++ conditional with the common type of both arms.
The oracle evaluates every sub-expression in exactly that type, in source
order, with IEEE arithmetic and no contraction (DaCe's default ``-ffast-math``
build makes the reference's own last bits compiler-dependent; strict IEEE is
the canonical representative).
"""

import ast
import json
import os
import re

import numpy as np

ITERATORS = ["i", "j", "k"]
JUNK_VAL = -100000  # stencilflow/stencil/_common.py:8

_NP = {
    "bool": np.bool_,
    "int32": np.int32,
    "int64": np.int64,
    "float32": np.float32,
    "float64": np.float64,
}
_WEAK = ("wfloat", "wint")
_RANK = {"bool": 0, "int32": 3, "int64": 5, "float32": 10, "float64": 11}


def _promote(a, b):
    """C++ usual arithmetic conversions on type names.  The names "wfloat" /
    "wint" (typing="nep50" only, see run_reference) stand for Python literals,
    which NumPy >= 2 treats as weakly typed: they adopt the other operand's
    type when that can hold them (float literal with an integer array:
    float64)."""
    wa, wb = a in _WEAK, b in _WEAK
    if wa or wb:
        if wa and wb:
            return "wfloat" if "wfloat" in (a, b) else "wint"
        weak, strong = (a, b) if wa else (b, a)
        if weak == "wint" or strong.startswith("float"):
            return "int64" if strong == "bool" else strong
        return "float64"
    fa, fb = a.startswith("float"), b.startswith("float")
    if fa or fb:
        if fa and fb:
            return a if _RANK[a] >= _RANK[b] else b
        return a if fa else b
    a = "int32" if _RANK[a] < 3 else a
    b = "int32" if _RANK[b] < 3 else b
    return a if _RANK[a] >= _RANK[b] else b


def _literal_type(v, typing="cxx"):
    if typing == "nep50":
        return "wfloat" if isinstance(v, float) else "wint"
    if isinstance(v, bool):
        return "int32"
    if isinstance(v, int):
        return "int32" if -2**31 <= v < 2**31 else "int64"
    if isinstance(v, float):
        return "float64"
    raise TypeError(v)


class _V:
    """A typed value: NumPy array (broadcastable to the domain) or scalar."""
    __slots__ = ("a", "t")

    def __init__(self, a, t):
        self.a = a
        self.t = t

    def cast(self, t):
        if t == self.t:
            return self.a
        if t in _WEAK:  # weak stays a Python number
            return float(self.a) if t == "wfloat" else self.a
        if self.t in _WEAK:
            return _NP[t](self.a)
        return np.asarray(self.a).astype(_NP[t])


# ---------------------------------------------------------------------------
# program loading
# ---------------------------------------------------------------------------


def load_program(path_or_dict):
    if isinstance(path_or_dict, dict):
        prog = json.loads(json.dumps(path_or_dict))
        prog.setdefault("path", os.getcwd())
        return prog
    with open(path_or_dict) as f:
        prog = json.load(f)
    prog["path"] = os.path.dirname(os.path.abspath(path_or_dict))
    return prog


def _own_iterators(prog):
    return ITERATORS[3 - len(prog["dimensions"]):]


def _input_dims(prog, name):
    desc = prog["inputs"][name]
    if "input_dims" in desc and desc["input_dims"] is not None:
        return list(desc["input_dims"])
    if "dimensions" in desc:
        return list(desc["dimensions"])
    # kernel_chain_graph.py:382-389
    return _own_iterators(prog)


def _dims_shape(prog, dims):
    own = _own_iterators(prog)
    return tuple(prog["dimensions"][own.index(d)] for d in dims)


def materialise_input(prog, name, prefix=None, override=None):
    """stencilflow/helper.py:162-217, shaped to the input's own dims."""
    desc = prog["inputs"][name]
    dt = _NP[desc["data_type"]]
    dims = _input_dims(prog, name)
    shape = _dims_shape(prog, dims)
    data = desc["data"] if override is None else override
    if isinstance(data, np.ndarray):
        return data.astype(dt, copy=False).reshape(shape)
    if isinstance(data, str):
        m = re.match(r"([^:]+):(.+)", data)
        if m and m.group(1) == "constant":
            val = float(m.group(2))
            if not dims:
                return dt(val)
            return np.full(shape, val, dtype=dt)
        if m and m.group(1) == "random":
            parts = [p for p in re.split(r"[,\s]+|\.\.", m.group(2)) if p]
            rng = np.random.default_rng(int(parts[2]) if len(parts) > 2 else 0)
            if not dims:
                return dt(rng.uniform(float(parts[0]), float(parts[1])))
            return rng.uniform(float(parts[0]), float(parts[1]),
                               shape).astype(dt)
        path = data
        for cand in (data, os.path.join(prefix or "", data),
                     os.path.join(prog["path"], data)):
            if os.path.isfile(cand):
                path = cand
                break
        else:
            raise FileNotFoundError(data)
        arr = np.genfromtxt(path, dt, delimiter=",") if path.endswith(".csv") \
            else np.fromfile(path, dt)
        return arr.reshape(shape)
    if not dims:
        return dt(data)
    return np.array(data, dtype=dt).reshape(shape)


# ---------------------------------------------------------------------------
# one operator on the whole domain
# ---------------------------------------------------------------------------


class _KernelEval(ast.NodeVisitor):
    """Evaluates one kernel's statements for all points at once.

    Per point the reference does (stencil/cpu.py:58-115): for each access with
    a non-zero offset ``v = bc if out_of_domain else field[p + offset]``, then
    the user's statements, then ``out[p] = <kernel name>``.
    """

    def __init__(self, prog, kname, fields, field_dims, field_types, scalars,
                 typing="cxx"):
        self.typing = typing
        self.prog = prog
        self.kname = kname
        self.kdesc = prog["program"][kname]
        self.fields = fields  # name -> ndarray over its own dims
        self.field_dims = field_dims
        self.field_types = field_types
        self.scalars = scalars  # name -> _V
        self.own = _own_iterators(prog)
        self.shape = tuple(prog["dimensions"])
        self.locals = {}
        self._padded = {}

    # -- accesses ---------------------------------------------------------
    def _offsets(self, node, dims):
        sl = node.slice
        elts = list(sl.elts) if isinstance(sl, ast.Tuple) else [sl]
        offs = {}
        for e in elts:
            if isinstance(e, ast.Name):
                offs[e.id] = 0
            else:  # it +/- c   (compute_graph_nodes.py:199-209)
                c = int(e.right.value)
                offs[e.left.id] = -c if isinstance(e.op, ast.Sub) else c
        if list(offs) != list(dims):
            raise ValueError("access {} does not match field dims {}".format(
                ast.unparse(node), dims))
        return [offs[d] for d in dims]

    def _bc(self, field):
        bc = self.kdesc["boundary_conditions"][field]
        kind = bc.get("type", bc.get("btype"))
        if kind == "constant":
            return bc["value"]
        if kind == "shrink":
            return JUNK_VAL  # stencil/cpu.py:89-93
        # "copy" raises NameError in the reference (stencil/cpu.py:87)
        raise ValueError(
            "Unsupported boundary condition type: {}".format(kind))

    def visit_Subscript(self, node):
        field = node.value.id
        dims = self.field_dims[field]
        offs = self._offsets(node, dims)
        arr = self.fields[field]
        ftype = self.field_types[field]
        if all(o == 0 for o in offs):  # stencil/cpu.py:82-84
            view, vtype = arr, ftype
        else:
            fill = self._bc(field)
            vtype = _promote(_literal_type(fill, self.typing), ftype)
            key = (field, vtype)
            if key not in self._padded:
                pads = [0] * len(dims)
                self._padded[key] = (pads, None)
            # (re)build the padded copy if this access reaches further
            need = [min(abs(o), n) for o, n in zip(offs, arr.shape)]
            pads, padded = self._padded[key]
            if padded is None or any(n > p for n, p in zip(need, pads)):
                pads = [max(n, p) for n, p in zip(need, pads)]
                padded = np.full([n + 2 * p for n, p in zip(arr.shape, pads)],
                                 fill,
                                 dtype=_NP[vtype])
                padded[tuple(slice(p, p + n)
                             for p, n in zip(pads, arr.shape))] = arr
                self._padded[key] = (pads, padded)
            if any(abs(o) >= n for o, n in zip(offs, arr.shape)):
                # every point reads outside the domain (e.g. kA[j,k-100])
                view = np.full(arr.shape, fill, dtype=_NP[vtype])
            else:
                view = padded[tuple(
                    slice(p + o, p + o + n)
                    for p, o, n in zip(pads, offs, arr.shape))]
        # broadcast a lower-dimensional field over the domain
        # (stencil/cpu.py:150-160: the memlet indexes only the field's dims)
        index = tuple(slice(None) if d in dims else None for d in self.own)
        return _V(view[index] if dims != self.own else view, vtype)

    # -- scalars ----------------------------------------------------------
    def visit_Constant(self, node):
        v = node.value
        t = _literal_type(v, self.typing)
        return _V(v if t in _WEAK else _NP[t](v), t)

    def visit_Name(self, node):
        if node.id in self.locals:
            return self.locals[node.id]
        if node.id in self.scalars:
            return self.scalars[node.id]
        raise ValueError("unknown name " + node.id)

    # -- operators --------------------------------------------------------
    def visit_BinOp(self, node):
        lhs, rhs = self.visit(node.left), self.visit(node.right)
        t = _promote(lhs.t, rhs.t)
        if t in _WEAK:  # two literals: plain Python arithmetic
            a, b = lhs.a, rhs.a
            r = {ast.Add: a + b, ast.Sub: a - b, ast.Mult: a * b,
                 ast.Div: a / b if b else float("nan")}[type(node.op)]
            return _V(r, "wfloat" if isinstance(r, float) else "wint")
        if isinstance(node.op, ast.Div) and not t.startswith("float"):
            t = "float64"  # Python true division
        a, b = lhs.cast(t), rhs.cast(t)
        with np.errstate(all="ignore"):
            if isinstance(node.op, ast.Add):
                r = np.add(a, b)
            elif isinstance(node.op, ast.Sub):
                r = np.subtract(a, b)
            elif isinstance(node.op, ast.Mult):
                r = np.multiply(a, b)
            elif isinstance(node.op, ast.Div):
                r = np.divide(a, b)
            else:
                raise TypeError(node.op)
        assert np.asarray(r).dtype == _NP[t], (np.asarray(r).dtype, t)
        return _V(r, t)

    def visit_UnaryOp(self, node):
        v = self.visit(node.operand)
        if isinstance(node.op, ast.UAdd):
            return v
        if v.t in _WEAK:
            return _V(-v.a, v.t)
        t = _promote(v.t, v.t)
        return _V(np.negative(v.cast(t)), t)

    def visit_Compare(self, node):
        lhs, rhs = self.visit(node.left), self.visit(node.comparators[0])
        t = _promote(lhs.t, rhs.t)
        if t in _WEAK:
            t = "float64" if t == "wfloat" else "int64"
        a, b = lhs.cast(t), rhs.cast(t)
        fn = {
            ast.Lt: np.less,
            ast.LtE: np.less_equal,
            ast.Gt: np.greater,
            ast.GtE: np.greater_equal,
            ast.Eq: np.equal,
            ast.NotEq: np.not_equal
        }[type(node.ops[0])]
        return _V(fn(a, b), "bool")

    def visit_BoolOp(self, node):
        a = self.visit(node.values[0])
        b = self.visit(node.values[1])
        fn = np.logical_and if isinstance(node.op, ast.And) else np.logical_or
        return _V(fn(np.asarray(a.a) != 0, np.asarray(b.a) != 0), "bool")

    def visit_IfExp(self, node):
        c = self.visit(node.test)
        x, y = self.visit(node.body), self.visit(node.orelse)
        t = _promote(x.t, y.t)
        if t in _WEAK:
            t = "float64" if t == "wfloat" else "int64"
        return _V(np.where(np.asarray(c.a) != 0, x.cast(t), y.cast(t)), t)

    def visit_Call(self, node):
        fn = node.func.id
        args = [self.visit(a) for a in node.args]
        t = args[0].t
        for a in args[1:]:
            t = _promote(t, a.t)
        if t in _WEAK:
            t = "float64" if t == "wfloat" else "int64"
        if fn in ("min", "max"):
            a, b = args[0].cast(t), args[1].cast(t)
            # dace::math::min/max are "(a < b) ? a : b" / "(a > b) ? a : b"
            r = np.where(a < b, a, b) if fn == "min" else np.where(a > b, a, b)
            return _V(r, t)
        if fn == "abs" and not t.startswith("float"):
            t = _promote(t, t)
            return _V(np.abs(args[0].cast(t)), t)
        if not t.startswith("float"):
            t = "float64"
        table = {
            "sin": np.sin,
            "cos": np.cos,
            "tan": np.tan,
            "sinh": np.sinh,
            "cosh": np.cosh,
            "sqrt": np.sqrt,
            "fabs": np.abs,
            "abs": np.abs,
            "exp": np.exp,
            "log": np.log
        }
        with np.errstate(all="ignore"):
            return _V(table[fn](args[0].cast(t)), t)

    def generic_visit(self, node):
        raise TypeError("unsupported syntax: " + type(node).__name__)

    # -- driver -----------------------------------------------------------
    def run(self):
        tree = ast.parse(self.kdesc["computation_string"])
        for stmt in tree.body:
            self.locals[stmt.targets[0].id] = self.visit(stmt.value)
        result = self.locals[self.kname]  # sdfg_generator.py:104-106
        out_t = self.kdesc["data_type"]
        out = np.empty(self.shape, dtype=_NP[out_t])
        with np.errstate(all="ignore"):
            out[...] = result.cast(out_t)  # broadcast + final conversion
        return out


def _reads(prog, kname):
    """Names a kernel reads (fields by subscript, scalars by bare name)."""
    tree = ast.parse(prog["program"][kname]["computation_string"])
    assigned = {s.targets[0].id for s in tree.body}
    names = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and n.id not in assigned \
                and n.id not in ITERATORS and n.id not in names:
            names.append(n.id)
    return names


def topological_kernels(prog):
    """Execution order: any topological order of the name-matched DAG
    (stencilflow/sdfg_generator.py:638-641); ties broken by program order."""
    kernels = list(prog["program"])
    deps = {
        k: [r for r in _reads(prog, k) if r in prog["program"] and r != k]
        for k in kernels
    }
    done, order = set(), []
    pending = list(kernels)
    while pending:
        progressed = False
        for k in list(pending):
            if all(d in done for d in deps[k]):
                order.append(k)
                done.add(k)
                pending.remove(k)
                progressed = True
        if not progressed:
            raise ValueError("Cycle detected: {}".format(pending))
    return order


def run_reference(program,
                  inputs=None,
                  input_directory=None,
                  return_all=False,
                  generate_input=False,
                  typing="cxx"):
    """Run the whole program the way the reference CPU program does.

    program : path to a program ``.json`` or an already-loaded dict
    inputs  : optional dict name -> ndarray / scalar overriding ``data``
    typing  : "cxx" (default) -- the TYPING CONTRACT of the module docstring,
              what the reference's DaCe-generated C++ does; "nep50" -- literals
              are weakly typed as in NumPy >= 2 scalar arithmetic, which is how
              the reference's *Simulator* evaluates an operator
              (stencilflow/kernel.py:700-709: ``data_type(eval_expr(...))`` on
              NumPy scalars and Python floats).  Used only to compare with
              vectors captured from that Simulator (tests/golden/
              simulator_vectors.json); identical to it as long as arithmetic
              among literals and boundary constants alone is exact.
    Returns dict output name -> ndarray of shape ``dimensions``
    (all kernels' fields when ``return_all``).
    """
    prog = load_program(program)
    inputs = inputs or {}
    own = _own_iterators(prog)
    fields, field_dims, field_types, scalars = {}, {}, {}, {}
    for name, desc in prog["inputs"].items():
        override = inputs.get(name)
        if override is None and generate_input:
            override = "constant:0.5"  # stencilflow/run_program.py:141-144
        val = materialise_input(prog, name, input_directory, override)
        dims = _input_dims(prog, name)
        if not dims:
            scalars[name] = _V(_NP[desc["data_type"]](val), desc["data_type"])
        else:
            fields[name] = val
            field_dims[name] = dims
            field_types[name] = desc["data_type"]
    for name, desc in prog.get("constants", {}).items():
        scalars[name] = _V(_NP[desc["data_type"]](desc["value"]),
                           desc["data_type"])
    results = {}
    for kname in topological_kernels(prog):
        ev = _KernelEval(prog, kname, fields, field_dims, field_types, scalars,
                         typing)
        out = ev.run()
        fields[kname] = out
        field_dims[kname] = own
        field_types[kname] = prog["program"][kname]["data_type"]
        results[kname] = out
    if return_all:
        return results
    return {name: results[name] for name in prog["outputs"]}


def arrays_match(reference, result, tolerance=1e-6):
    """|ref - res| <= tol * max(|ref|, |res|) everywhere, no NaNs."""
    reference, result = np.asarray(reference), np.asarray(result)
    if reference.shape != result.shape:
        return False
    if np.isnan(reference).any() or np.isnan(result).any():
        return False
    scale = np.maximum(np.abs(reference), np.abs(result))
    return bool(np.all(np.abs(reference - result) <= tolerance * scale))


def max_rel_err(reference, result):
    reference = np.asarray(reference, dtype=np.float64)
    result = np.asarray(result, dtype=np.float64)
    scale = np.maximum(np.abs(reference), np.abs(result))
    with np.errstate(all="ignore"):
        rel = np.where(scale > 0, np.abs(reference - result) / scale, 0.0)
    return float(rel.max()) if rel.size else 0.0
