"""Front end for the stencil expression DSL.

A kernel's ``computation_string`` is a short list of Python-syntax assignments
(``;`` or newline separated); the statement assigning the kernel's own name
defines the operator's value at the centre point.  Grammar accepted (same as
the reference's ``ComputeGraph``, stencilflow/compute_graph.py:81-110,203-326
and compute_graph_nodes.py:189-238):

* ``+ - * /``, unary ``-``, numbers, names (scalar inputs, program constants,
  earlier locals);
* field accesses ``f[it, it+c, it-c]`` with ``it`` in ``i, j, k``;
* calls with one or two arguments out of the table in
  stencilflow/compute_graph.config:3-19 (``sin cos tan sinh cosh sqrt min max
  fabs``) plus ``exp log abs``;
* ``a if c else b``, comparisons ``< <= > >= == !=``, ``and`` / ``or`` with two
  operands.

The tree built here is typed with C++ arithmetic-conversion rules
(``dtypes.promote``) because the reference CPU path is C++ emitted from this
very text (stencilflow/stencil/cpu.py:46-115): a Python ``float`` literal is a
``double``, an ``int`` literal an ``int``, a float32 field load a ``float``.
"""

import ast

from . import dtypes
from .helper import ITERATORS

JUNK_VAL = -100000  # reference stencilflow/stencil/_common.py:8 ("shrink" BC)

UNARY_CALLS = ("sin", "cos", "tan", "sinh", "cosh", "sqrt", "fabs", "abs",
               "exp", "log")
BINARY_CALLS = ("min", "max")

_BINOPS = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}
_CMPOPS = {
    ast.Lt: "<",
    ast.LtE: "<=",
    ast.Gt: ">",
    ast.GtE: ">=",
    ast.Eq: "==",
    ast.NotEq: "!="
}


class Node:
    dtype = None


class Const(Node):
    def __init__(self, value):
        self.value = value
        self.dtype = dtypes.literal_type(value)


class Ref(Node):
    """A scalar: 0-D program input, program constant or kernel-local."""

    def __init__(self, name, dtype, kind):
        self.name = name
        self.dtype = dtype
        self.kind = kind  # "scalar" | "constant" | "local"


class Load(Node):
    """``field[...]`` with a 3-D relative index (``None`` for absent dims)."""

    def __init__(self, field, index, dtype):
        self.field = field
        self.index = tuple(index)
        self.dtype = dtype


class Bin(Node):
    def __init__(self, op, lhs, rhs):
        self.op, self.lhs, self.rhs = op, lhs, rhs
        self.dtype = dtypes.promote(lhs.dtype, rhs.dtype)
        if op == "/" and not self.dtype.is_float:
            # Python "/" is true division; integer operands become double
            self.dtype = dtypes.float64


class Neg(Node):
    def __init__(self, operand):
        self.operand = operand
        self.dtype = dtypes.promote(operand.dtype, operand.dtype)


class Cmp(Node):
    def __init__(self, op, lhs, rhs):
        self.op, self.lhs, self.rhs = op, lhs, rhs
        self.dtype = dtypes.bool_


class BoolOp(Node):
    def __init__(self, op, lhs, rhs):
        self.op, self.lhs, self.rhs = op, lhs, rhs  # op: "&&" | "||"
        self.dtype = dtypes.bool_


class Select(Node):
    def __init__(self, cond, if_true, if_false):
        self.cond, self.if_true, self.if_false = cond, if_true, if_false
        self.dtype = dtypes.promote(if_true.dtype, if_false.dtype)


class Call(Node):
    def __init__(self, func, args):
        self.func, self.args = func, args
        t = args[0].dtype
        for a in args[1:]:
            t = dtypes.promote(t, a.dtype)
        if func in UNARY_CALLS and func not in ("abs", "fabs") \
                and not t.is_float:
            t = dtypes.float64
        if func == "fabs" and not t.is_float:
            t = dtypes.float64
        self.dtype = t


def parse_index(subscript, field_dims):
    """``f[i+1, k]`` -> ``(1, None, 0)`` for a field defined over ``field_dims``.

    Each element must be ``it``, ``it + c`` or ``it - c``
    (reference compute_graph_nodes.py:189-223) and name the field's dimensions
    in order.
    """
    sl = subscript.slice
    if isinstance(sl, ast.Index):  # Python < 3.9 layout
        sl = sl.value
    elements = list(sl.elts) if isinstance(sl, ast.Tuple) else [sl]
    offsets = {}
    order = []
    for elem in elements:
        if isinstance(elem, ast.Name):
            it, off = elem.id, 0
        elif (isinstance(elem, ast.BinOp)
              and isinstance(elem.op, (ast.Add, ast.Sub))
              and isinstance(elem.left, ast.Name)
              and isinstance(elem.right, ast.Constant)
              and isinstance(elem.right.value, int)):
            it = elem.left.id
            off = elem.right.value
            if isinstance(elem.op, ast.Sub):
                off = -off
        else:
            raise TypeError("Unrecognized offset: {}".format(
                ast.unparse(elem)))
        if it not in ITERATORS:
            raise TypeError("Unknown iterator '{}' in {}".format(
                it, ast.unparse(subscript)))
        offsets[it] = off
        order.append(it)
    if order != list(field_dims):
        raise ValueError(
            "Access {} does not index the dimensions {} of the field".format(
                ast.unparse(subscript), list(field_dims)))
    return tuple(offsets.get(it) for it in ITERATORS)


class KernelExpr:
    """Parsed, typed statements of one stencil operator.

    Attributes
    ----------
    statements : list of (target name, Node)
    accesses   : dict field -> list of distinct 3-D index tuples, in first-use
                 order (what the reference keeps in ``ComputeGraph.accesses``,
                 compute_graph.py:125-144)
    scalars    : names of 0-D inputs / constants referenced
    """

    def __init__(self, name, computation_string, field_info, scalar_info,
                 boundary_conditions):
        """
        field_info  : dict field -> (dims list e.g. ["j","k"], dtype)
        scalar_info : dict name  -> (dtype, kind)
        boundary_conditions : the kernel's BC dict (field -> {type, value})
        """
        self.name = name
        self.text = computation_string
        self.field_info = field_info
        self.scalar_info = scalar_info
        self.boundary_conditions = boundary_conditions
        self.accesses = {}
        self.scalars = []
        self.locals = {}
        self.statements = []
        tree = ast.parse(computation_string)
        for stmt in tree.body:
            if not isinstance(stmt, ast.Assign) or len(stmt.targets) != 1 \
                    or not isinstance(stmt.targets[0], ast.Name):
                raise ValueError(
                    "Kernel '{}': only 'name = expression' statements are "
                    "supported, got: {}".format(name, ast.unparse(stmt)))
            target = stmt.targets[0].id
            node = self._walk(stmt.value)
            self.locals[target] = node.dtype
            self.statements.append((target, node))
        if name not in self.locals:
            raise ValueError(
                "Kernel '{}' never assigns its own name".format(name))

    # -- typing of one field access ---------------------------------------
    def access_dtype(self, field, index):
        """Type of the value an access yields after the boundary select.

        An access with a non-zero offset becomes ``bc if oob else load`` in the
        reference's tasklet (stencil/cpu.py:71-102), a C++ conditional whose
        type is the common type of the BC literal and the field's type.
        """
        _, fdtype = self.field_info[field]
        if all(o in (0, None) for o in index):
            return fdtype
        bc = self.boundary_conditions.get(field)
        if bc is None:
            raise ValueError(
                "Kernel '{}': no boundary condition for field '{}'".format(
                    self.name, field))
        btype = bc.get("type", bc.get("btype"))
        if btype == "constant":
            return dtypes.promote(dtypes.literal_type(bc["value"]), fdtype)
        if btype == "shrink":
            return dtypes.promote(dtypes.literal_type(JUNK_VAL), fdtype)
        if btype == "copy":
            return fdtype
        raise ValueError(
            "Unsupported boundary condition type: {}".format(btype))

    def _walk(self, node):
        if isinstance(node, ast.Constant):
            if isinstance(node.value, (int, float)):
                return Const(node.value)
            raise TypeError("Unsupported literal {!r}".format(node.value))
        if isinstance(node, ast.Name):
            nm = node.id
            if nm in self.locals:
                return Ref(nm, self.locals[nm], "local")
            if nm in self.scalar_info:
                dtype, kind = self.scalar_info[nm]
                if nm not in self.scalars:
                    self.scalars.append(nm)
                return Ref(nm, dtype, kind)
            if nm in self.field_info:
                raise ValueError(
                    "Kernel '{}': field '{}' used without an index".format(
                        self.name, nm))
            raise ValueError("Kernel '{}': unknown name '{}'".format(
                self.name, nm))
        if isinstance(node, ast.Subscript):
            if not isinstance(node.value, ast.Name):
                raise TypeError("Only subscripts of variables are supported")
            field = node.value.id
            if field not in self.field_info:
                raise ValueError("Kernel '{}': unknown field '{}'".format(
                    self.name, field))
            dims, _ = self.field_info[field]
            index = parse_index(node, dims)
            lst = self.accesses.setdefault(field, [])
            if index not in lst:
                lst.append(index)
            return Load(field, index, self.access_dtype(field, index))
        if isinstance(node, ast.BinOp):
            if type(node.op) not in _BINOPS:
                raise TypeError("Unsupported operator {}".format(
                    type(node.op).__name__))
            return Bin(_BINOPS[type(node.op)], self._walk(node.left),
                       self._walk(node.right))
        if isinstance(node, ast.UnaryOp):
            if isinstance(node.op, ast.USub):
                return Neg(self._walk(node.operand))
            if isinstance(node.op, ast.UAdd):
                return self._walk(node.operand)
            raise TypeError("Unsupported unary operator")
        if isinstance(node, ast.Compare):
            if len(node.ops) != 1 or type(node.ops[0]) not in _CMPOPS:
                raise TypeError("Unsupported comparison")
            return Cmp(_CMPOPS[type(node.ops[0])], self._walk(node.left),
                       self._walk(node.comparators[0]))
        if isinstance(node, ast.BoolOp):
            if len(node.values) != 2:
                raise NotImplementedError(
                    "Boolean operators take exactly two operands")
            op = "&&" if isinstance(node.op, ast.And) else "||"
            return BoolOp(op, self._walk(node.values[0]),
                          self._walk(node.values[1]))
        if isinstance(node, ast.IfExp):
            return Select(self._walk(node.test), self._walk(node.body),
                          self._walk(node.orelse))
        if isinstance(node, ast.Call):
            if not isinstance(node.func, ast.Name):
                raise TypeError("Unsupported call")
            fn = node.func.id
            nargs = len(node.args)
            if nargs > 2:
                raise NotImplementedError(
                    "Calls with more than two arguments are not supported")
            if (fn in UNARY_CALLS and nargs == 1) or (fn in BINARY_CALLS
                                                      and nargs == 2):
                return Call(fn, [self._walk(a) for a in node.args])
            raise ValueError("Unsupported function '{}' with {} args".format(
                fn, nargs))
        raise Exception("Unknown AST type {}".format(type(node)))


# ---------------------------------------------------------------------------
# C emission (shared by the HIP kernel generator; the oracle has its own)
# ---------------------------------------------------------------------------


def access_var(field, index):
    """Local name for ``field[index]`` in the style of the reference's
    ``SubscriptConverter`` (stencil/subscript_converter.py:12-29):
    ``a[-1,0,1]`` -> ``a_m1_0_1``; absent dims are dropped."""
    parts = [("m" + str(-o)) if o < 0 else str(o) for o in index
             if o is not None]
    return field + "_" + "_".join(parts) if parts else field + "_s"


def c_literal(value):
    if isinstance(value, bool):
        return "1" if value else "0"
    if isinstance(value, int):
        return str(value) if -2**31 <= value < 2**31 else str(value) + "LL"
    text = repr(float(value))
    if text in ("inf", "-inf", "nan"):
        return {"inf": "INFINITY", "-inf": "(-INFINITY)", "nan": "NAN"}[text]
    if "." not in text and "e" not in text:
        text += ".0"
    return text


def _c_call(func, args, dtype):
    single = dtype == dtypes.float32
    if func in ("min", "max"):
        op = "<" if func == "min" else ">"
        a, b = args
        return "(({a}) {op} ({b}) ? ({a}) : ({b}))".format(a=a, b=b, op=op)
    if func == "abs" and not dtype.is_float:
        return "(({a}) < 0 ? -({a}) : ({a}))".format(a=args[0])
    base = {"abs": "fabs"}.get(func, func)
    cast = "" if dtype.is_float else "(double)"
    return "{}{}({}{})".format(base, "f" if single else "", cast, args[0])


def to_c(node, rename=None):
    """Fully parenthesised C expression for ``node``.  Promotions are left to
    the C compiler except where Python and C differ (``/`` on integers)."""
    rename = rename or {}
    if isinstance(node, Const):
        return c_literal(node.value)
    if isinstance(node, Ref):
        return rename.get(node.name, node.name)
    if isinstance(node, Load):
        return rename.get((node.field, node.index),
                          access_var(node.field, node.index))
    if isinstance(node, Bin):
        lhs, rhs = to_c(node.lhs, rename), to_c(node.rhs, rename)
        if node.op == "/" and not dtypes.promote(node.lhs.dtype,
                                                 node.rhs.dtype).is_float:
            lhs = "(double)" + lhs
        return "({} {} {})".format(lhs, node.op, rhs)
    if isinstance(node, Neg):
        return "(-{})".format(to_c(node.operand, rename))
    if isinstance(node, (Cmp, BoolOp)):
        return "({} {} {})".format(to_c(node.lhs, rename), node.op,
                                   to_c(node.rhs, rename))
    if isinstance(node, Select):
        return "({} ? {} : {})".format(to_c(node.cond, rename),
                                       to_c(node.if_true, rename),
                                       to_c(node.if_false, rename))
    if isinstance(node, Call):
        return _c_call(node.func, [to_c(a, rename) for a in node.args],
                       node.dtype)
    raise TypeError(type(node))


def count_ops(node, counts=None):
    """Arithmetic operation census of a typed tree (informational)."""
    counts = {} if counts is None else counts
    if isinstance(node, Bin):
        counts[node.op] = counts.get(node.op, 0) + 1
        count_ops(node.lhs, counts)
        count_ops(node.rhs, counts)
    elif isinstance(node, Neg):
        counts["neg"] = counts.get("neg", 0) + 1
        count_ops(node.operand, counts)
    elif isinstance(node, (Cmp, BoolOp)):
        counts["cmp"] = counts.get("cmp", 0) + 1
        count_ops(node.lhs, counts)
        count_ops(node.rhs, counts)
    elif isinstance(node, Select):
        counts["sel"] = counts.get("sel", 0) + 1
        for c in (node.cond, node.if_true, node.if_false):
            count_ops(c, counts)
    elif isinstance(node, Call):
        counts[node.func] = counts.get(node.func, 0) + 1
        for a in node.args:
            count_ops(a, counts)
    return counts
