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
