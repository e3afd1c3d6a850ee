"""Synthetic stencil programs in the reference's JSON format.

The generators follow the conventions of the reference's workload generator
(bin/synthesize.py:119-125,167,206-210: input ``a``, stages ``b0..b<N-1>``, one
output = the last stage, constant boundary conditions) and produce the
configurations BASELINE.json names:

* ``jacobi3d(n, stages)``  — C1/C3/C4: 6-point ``0.16666666 * (...)`` chain, f32
* ``jacobi2d(n, stages)``  — C2: 4-point ``0.25 * (...)`` chain, f32
* ``diffusion_advection_laplacian(n)`` — C5: three different operators, f64
"""

import json
import os


def _chain(dimensions, stages, make_expr, data_type, bc_value, data,
           vectorization):
    program = {
        "inputs": {
            "a": {
                "data": data,
                "data_type": data_type
            }
        },
        "outputs": ["b{}".format(stages - 1)],
        "dimensions": list(dimensions),
        "vectorization": vectorization,
        "program": {}
    }
    prev = "a"
    for s in range(stages):
        name = "b{}".format(s)
        program["program"][name] = {
            "computation_string": "{} = {}".format(name, make_expr(prev)),
            "boundary_conditions": {
                prev: {
                    "type": "constant",
                    "value": bc_value
                }
            },
            "data_type": data_type
        }
        prev = name
    return program


def jacobi3d(dimensions=(512, 512, 512),
             stages=1000,
             data_type="float32",
             coefficient="0.16666666",
             bc_value=0.0,
             data="constant:1.0",
             vectorization=1):
    """The operator of test/stencils/jacobi3d_32x32x32_8itr.json, chained."""
    def expr(f):
        return ("{c} * ({f}[i-1,j,k] + {f}[i+1,j,k] + {f}[i,j-1,k] + "
                "{f}[i,j+1,k] + {f}[i,j,k-1] + {f}[i,j,k+1])").format(
                    c=coefficient, f=f)

    return _chain(dimensions, stages, expr, data_type, bc_value, data,
                  vectorization)


def jacobi2d(dimensions=(4096, 4096),
             stages=1000,
             data_type="float32",
             coefficient="0.25",
             bc_value=0.0,
             data="constant:1.0",
             vectorization=1):
    """The operator of test/stencils/jacobi2d_128x128.json, chained."""
    def expr(f):
        return ("{c} * ({f}[j-1,k] + {f}[j+1,k] + {f}[j,k-1] + {f}[j,k+1])"
                ).format(c=coefficient, f=f)

    return _chain(dimensions, stages, expr, data_type, bc_value, data,
                  vectorization)


def diffusion_advection_laplacian(dimensions=(512, 512, 512),
                                  data_type="float64",
                                  data="constant:1.0",
                                  repeats=1):
    """C5: diffusion (7-point, scalar coefficients) -> first-order upwind
    advection -> 7-point Laplacian; builder-defined, see SURVEY.md §8d."""
    scalars = {
        "c0": 0.4, "c1": 0.1, "c2": 0.1, "c3": 0.1, "c4": 0.1, "c5": 0.1,
        "c6": 0.1, "cx": 0.2, "cy": 0.15, "cz": 0.1
    }
    program = {
        "inputs": {
            "a": {
                "data": data,
                "data_type": data_type
            }
        },
        "outputs": [],
        "dimensions": list(dimensions),
        "program": {}
    }
    for name, value in scalars.items():
        program["inputs"][name] = {
            "data": value,
            "data_type": data_type,
            "input_dims": []
        }
    prev = "a"
    for r in range(repeats):
        sfx = "" if repeats == 1 else str(r)
        diff, adv, lap = "diff" + sfx, "adv" + sfx, "lap" + sfx
        program["program"][diff] = {
            "computation_string":
            ("{d} = c0*{f}[i,j,k] + c1*{f}[i-1,j,k] + c2*{f}[i+1,j,k] + "
             "c3*{f}[i,j-1,k] + c4*{f}[i,j+1,k] + c5*{f}[i,j,k-1] + "
             "c6*{f}[i,j,k+1]").format(d=diff, f=prev),
            "boundary_conditions": {prev: {"type": "constant", "value": 0.0}},
            "data_type": data_type
        }
        program["program"][adv] = {
            "computation_string":
            ("{a} = {f}[i,j,k] - cx*({f}[i,j,k] - {f}[i-1,j,k]) - "
             "cy*({f}[i,j,k] - {f}[i,j-1,k]) - cz*({f}[i,j,k] - {f}[i,j,k-1])"
             ).format(a=adv, f=diff),
            "boundary_conditions": {diff: {"type": "constant", "value": 0.0}},
            "data_type": data_type
        }
        program["program"][lap] = {
            "computation_string":
            ("{l} = {f}[i-1,j,k] + {f}[i+1,j,k] + {f}[i,j-1,k] + {f}[i,j+1,k]"
             " + {f}[i,j,k-1] + {f}[i,j,k+1] - 6.0*{f}[i,j,k]").format(l=lap,
                                                                      f=adv),
            "boundary_conditions": {adv: {"type": "constant", "value": 0.0}},
            "data_type": data_type
        }
        prev = lap
    program["outputs"] = [prev]
    return program


def write_program(program, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(program, f, indent=1)
    return path


def _offset_text(iterator, value):
    if value > 0:
        return "{}+{}".format(iterator, value)
    if value < 0:
        return "{}-{}".format(iterator, -value)
    return iterator


def synthesize(data_type,
               num_stages,
               num_fields_spatial,
               size_x,
               size_y,
               size_z,
               extent_x,
               extent_y,
               extent_z,
               fork_frequency=0.0,
               fork_length_left=2,
               fork_length_right=2,
               stencil_shape="cross",
               vectorize=1):
    """Synthetic program generator with the parameters and output conventions of
    the reference's bin/synthesize.py (:34-294): a chain of `num_stages`
    stencils `b0..` over input `a`, optional extra input fields per stage,
    optional fork/join sections, shapes cross | box | diffusion | hotspot.
    Returns (program dict, canonical file name)."""
    sizes = [size_x, size_y, size_z]
    extents = [extent_x, extent_y, extent_z]
    shape = [s for s in sizes if s > 0]
    nd = len(shape)
    iterators = ["i", "j", "k"][3 - nd:]

    per_dim = []
    for size, extent in list(zip(sizes, extents))[:nd]:
        if extent == 0:
            per_dim.append([0])
        elif stencil_shape == "box":
            per_dim.append(list(range(-extent, extent + 1)))
        else:
            per_dim.append([o for o in range(-extent, extent + 1) if o != 0])
    offsets = []
    if stencil_shape == "box":
        def rec(d, cur):
            if d == nd:
                offsets.append(tuple(cur))
                return
            for o in per_dim[d]:
                rec(d + 1, cur + [o])
        rec(0, [])
    else:
        if stencil_shape in ("diffusion", "hotspot"):
            offsets.append(tuple([0] * nd))
        for d in range(nd):
            for o in per_dim[d]:
                offsets.append(tuple(o if e == d else 0 for e in range(nd)))
    index_texts = [", ".join(_offset_text(it, o) for it, o in zip(iterators, off))
                   for off in offsets]

    program = {
        "inputs": {"a": {"data": "constant:1", "data_type": data_type,
                         "input_dims": list(iterators)}},
        "outputs": [],
        "program": {},
        "dimensions": shape,
        "vectorization": vectorize,
    }
    state = {"fields": 1, "credit": 0.0}

    def expression(name, fields):
        if stencil_shape == "hotspot":
            f = fields[0]
            power = "power" if state["fields"] - 1 == 0 else "power{}".format(state["fields"] - 1)
            if nd == 3:
                return ("{n} = cc * {f}[i, j, k] + cn * {f}[i, j-1, k] + cs * {f}[i, j+1, k] + "
                        "cw * {f}[i, j, k-1] + ce * {f}[i, j, k+1] + ca * {f}[i-1, j, k] + "
                        "cb * {f}[i+1, j, k] + sdc * {p}[i, j, k] + ca * amb").format(n=name, f=f, p=power)
            if nd == 2:
                return ("{n} = {f}[j, k] + sdc * ({p}[j, k] + "
                        "({f}[j-1, k] + {f}[j+1, k] - 2.0 * {f}[j, k]) * r_y + "
                        "({f}[j, k-1] + {f}[j, k+1] - 2.0 * {f}[j, k]) * r_x + "
                        "(amb - {f}[j, k]) * r_z)").format(n=name, f=f, p=power)
            raise ValueError("Unsupported number of indices for hotspot.")
        operands = ["{}[{}]".format(f, ix) for f in fields for ix in index_texts]
        if stencil_shape == "diffusion":
            return "{} = {}".format(name, " + ".join(
                "c{}*{}".format(i, o) for i, o in enumerate(operands)))
        return "{} = {}*({})".format(name, 1 / len(operands), " + ".join(operands))

    def add_stencil(sources, name):
        extra = []
        state["credit"] += num_fields_spatial
        if state["credit"] >= 1:
            while state["credit"] >= 1:
                prefix = "power" if stencil_shape == "hotspot" else "a"
                field = "{}{}".format(prefix, state["fields"])
                extra.append(field)
                program["inputs"][field] = {"data": "constant:0.5", "data_type": data_type}
                state["fields"] += 1
                state["credit"] -= 1
        elif stencil_shape == "hotspot":
            extra.append("power" if state["fields"] - 1 < 1 else "power{}".format(state["fields"] - 1))
        fields = list(sources) + extra
        program["program"][name] = {
            "data_type": data_type,
            "boundary_conditions": {f: {"type": "constant", "value": 0} for f in fields},
            "computation_string": expression(name, fields),
        }

    previous, joins, fork_credit, name = "a", [], 0.0, "a"
    for stage in range(num_stages):
        name = "b{}".format(stage)
        add_stencil(joins if joins else [previous], name)
        joins = []
        fork_credit += fork_frequency
        if stage < num_stages - 1 and fork_credit >= 1:
            for tag, length in (("a", fork_length_left), ("b", fork_length_right)):
                tail = name
                for i in range(length):
                    branch = "{}{}{}".format(name, tag, i)
                    add_stencil([tail], branch)
                    tail = branch
                joins.append(tail)
            fork_credit = 0.0
        previous = name

    scalars = []
    if stencil_shape == "hotspot":
        program["inputs"]["power"] = {"data": "constant:0.5", "data_type": data_type}
        scalars = (["sdc", "r_x", "r_y", "r_z", "amb"] if nd == 2 else
                   ["cc", "cn", "cs", "cw", "ce", "ca", "cb", "sdc", "amb"])
        if nd not in (2, 3):
            raise NotImplementedError
    elif stencil_shape == "diffusion":
        scalars = ["c{}".format(i) for i in range(len(index_texts))]
    for s in scalars:
        program["inputs"][s] = {"data": "constant:0.5", "data_type": data_type, "input_dims": []}
    program["outputs"].append(name)

    args = [data_type, num_stages, num_fields_spatial, size_x, size_y, size_z, extent_x, extent_y,
            extent_z, fork_frequency, fork_length_left, fork_length_right, stencil_shape, vectorize]
    filename = "_".join(map(str, args)).replace(".", "p") + ".json"
    return program, filename
