"""Lowering: ``KernelChainGraph`` -> SFIR, the text the native backend consumes.

This is the GPU counterpart of the reference's ``generate_sdfg`` /
``generate_reference`` (stencilflow/sdfg_generator.py:219-577, 580-677) and of
``_generate_stencil`` (sdfg_generator.py:68-176): per operator it records the
accesses with their relative offsets, the boundary condition of every access,
the typed statements and the output type; per program the fields (inputs,
intermediates, outputs), scalars and the execution order.  Where the reference
hands this record to DaCe, which emits and compiles HLS/C++, this backend hands
it to ``libsf_hip.so`` (``sf_plan_create``), which emits and compiles HIP.

SFIR grammar (one record per line, blank-separated)::

    sfir 1
    program <name>
    dims <nd> <n0> .. <n(nd-1)>               own dimensions, slowest first
    scalar <name> <dtype> input               0-D program input (run-time value)
    scalar <name> <dtype> const <literal>     program constant
    field <name> <dtype> <mask> <role>        mask e.g. 101 over own dims;
                                              role input | temp | output
    kernel <name> <dtype>                     in execution (topological) order
    acc <var> <field> <vtype> <bc> <bcval> <o0> .. <o(nd-1)>
                                              one distinct access; offset "x" for
                                              a dim the field lacks; bc none |
                                              constant | shrink | copy
    use <scalar>
    let <ctype> <var> = <C expression>        statements, in source order
    ret <var>
    end
"""


from . import dtypes
from .expr import JUNK_VAL, access_var, c_literal, to_c
from .helper import ITERATORS
from .kernel_chain_graph import Kernel

_SHORT = {
    "float32": "f32",
    "float64": "f64",
    "int32": "i32",
    "int64": "i64",
    "bool": "i32",
}


def _short(dtype):
    try:
        return _SHORT[dtype.name]
    except KeyError:
        raise ValueError("Data type {} is not supported by the HIP backend".
                         format(dtype.name))


def _sanitize(name):
    ok = name.replace("_", "a").isalnum() and not name[0].isdigit()
    if not ok:
        raise ValueError("'{}' is not a valid identifier".format(name))
    return name


def lower(chain):
    """Return the SFIR text of ``chain`` (a ``KernelChainGraph``)."""
    nd = chain.kernel_dimensions
    own = ITERATORS[len(ITERATORS) - nd:]
    shape = chain.dimensions[len(chain.dimensions) - nd:]
    # reference sdfg_generator.py:43-45
    if chain.vectorization > 1 and shape[-1] % chain.vectorization != 0:
        raise ValueError("Shape not divisible by vectorization width")

    lines = ["sfir 1", "program " + _sanitize(chain.name.replace(".", "_"))]
    lines.append("dims {} {}".format(nd, " ".join(str(int(s)) for s in shape)))

    # scalars: 0-D inputs are run-time symbols (sdfg_generator.py:623-624),
    # program constants are baked (sdfg_generator.py:586-587)
    for name, (dtype, kind) in chain.scalar_info.items():
        if kind == "constant":
            value = chain.constants[name]["value"]
            lines.append("scalar {} {} const {}".format(
                _sanitize(name), _short(dtype), c_literal(dtype.type(value).item())))
        else:
            lines.append("scalar {} {} input".format(_sanitize(name),
                                                     _short(dtype)))

    def mask_of(dims):
        return "".join("1" if it in dims else "0" for it in own)

    # inputs keep their own dimensionality (sdfg_generator.py:601-612)
    for name, node in chain.input_nodes.items():
        if name in chain.scalar_info:
            continue
        dims, dtype = chain.field_info[name]
        if any(d not in own for d in dims):
            raise ValueError("Input '{}' uses dimensions {} outside the "
                             "program's {}".format(name, dims, own))
        lines.append("field {} {} {} input".format(_sanitize(name),
                                                   _short(dtype),
                                                   mask_of(dims)))

    kernels = chain.topological_kernels()
    writers = {}
    for k in kernels:
        role = "output" if k.name in chain.output_nodes else "temp"
        if role == "temp" and not any(
                isinstance(s, Kernel) for s in chain.graph.successors(k)):
            # reference warns / raises for orphans (sdfg_generator.py:475-481)
            raise ValueError("Orphan stencil: " + k.name)
        if k.name in writers:  # sdfg_generator.py:434-435
            raise RuntimeError("Multiple writers for " + k.name)
        writers[k.name] = k
        lines.append("field {} {} {} {}".format(_sanitize(k.name),
                                                _short(k.data_type),
                                                "1" * nd, role))

    for k in kernels:
        ex = k.expr
        lines.append("kernel {} {}".format(k.name, _short(k.data_type)))
        for field, indices in ex.accesses.items():
            dims, _ = chain.field_info[field]
            bc = k.boundary_conditions.get(field, {})
            kind = bc.get("type", bc.get("btype", "none"))
            for index in indices:
                centre = all(o in (0, None) for o in index)
                offs = []
                for it, o in zip(ITERATORS, index):
                    if it not in own:
                        continue
                    offs.append("x" if o is None else str(o))
                if centre:
                    bkind, bval = "none", "-"
                elif kind == "constant":
                    bkind, bval = "constant", c_literal(bc["value"])
                elif kind == "shrink":
                    bkind, bval = "shrink", c_literal(JUNK_VAL)
                elif kind == "copy":
                    bkind, bval = "copy", "-"
                else:
                    raise ValueError(
                        "Unsupported boundary condition type: {}".format(kind))
                lines.append("acc {} {} {} {} {} {}".format(
                    access_var(field, index), field,
                    _short(ex.access_dtype(field, index)), bkind, bval,
                    " ".join(offs)))
        for s in ex.scalars:
            lines.append("use " + s)
        for target, node in ex.statements:
            ctype = node.dtype.ctype if node.dtype != dtypes.bool_ else "int"
            lines.append("let {} {} = {}".format(ctype, _sanitize(target),
                                                 to_c(node)))
        lines.append("ret " + k.name)
        lines.append("end")
    return "\n".join(lines) + "\n"
