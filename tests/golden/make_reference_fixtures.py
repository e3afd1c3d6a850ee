#!/usr/bin/env python3
"""Generates tests/golden/reference_structure.json and simulator12_expected.json
by importing the REFERENCE implementation from /root/reference.

Runs only in the build container (the reference tree does not exist on the GPU
box; the tests read the committed JSON, never this script's imports).

The reference's analysis layer needs two harness-side shims to import under
Python 3.10 without DaCe:
  * a stub ``dace`` module whose ``dace.dtypes.<name>`` objects expose the
    ``.type`` / ``.bytes`` / ``__call__`` surface the analysis layer touches;
  * an ``ast.parse`` wrapper that re-inserts the ``ast.Index`` node Python <= 3.8
    produced (the reference reads ``node.slice.value``).
Nothing of the reference's source is copied: the fixture holds only data the
reference computes (node names, edges, access offsets, dimensions, strings it
re-emits, one simulator output vector).
"""

import ast
import json
import os
import sys
import types
from unittest import mock

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    class _Lenient(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            m = mock.MagicMock(name="dace." + item)
            setattr(self, item, m)
            return m

    dace = _Lenient("dace")
    dtypes = _Lenient("dace.dtypes")

    class typeclass:
        def __init__(self, name):
            self.name = name
            self.type = getattr(np, name)
            self.bytes = np.dtype(self.type).itemsize

        def __call__(self, v):
            return self.type(v)

        def __repr__(self):
            return self.name

    dtypes.typeclass = typeclass
    for n in ("float32", "float64", "int32", "int64"):
        setattr(dtypes, n, typeclass(n))
    dace.dtypes = dtypes
    sys.modules["dace"] = dace
    sys.modules["dace.dtypes"] = dtypes
    # any "import dace.<anything>" resolves to a lenient mock package
    import importlib.abc
    import importlib.machinery

    class _Loader(importlib.abc.Loader):
        def create_module(self, spec):
            m = _Lenient(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass

    class _Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, fullname, path, target=None):
            if fullname.startswith("dace.") and fullname != "dace.dtypes":
                return importlib.machinery.ModuleSpec(fullname, _Loader(),
                                                      is_package=True)
            return None

    dace.__path__ = []
    sys.meta_path.insert(0, _Finder())

    real_parse = ast.parse

    class _Reindex(ast.NodeTransformer):
        def visit_Subscript(self, node):
            self.generic_visit(node)
            if not isinstance(node.slice, ast.Index):
                idx = ast.Index()
                idx.value = node.slice
                idx._fields = ("value", )
                node.slice = idx
            return node

    if not hasattr(ast, "Index") or True:

        class Index(ast.AST):
            _fields = ("value", )

        ast.Index = Index

    def parse(source, *a, **k):
        return _Reindex().visit(real_parse(source, *a, **k))

    ast.parse = parse
    # ast.Num / node.n compatibility is still provided by Python 3.10


def main():
    install_shims()
    sys.path.insert(0, REFERENCE)
    import stencilflow  # noqa: F401  (the reference)
    from stencilflow.kernel_chain_graph import KernelChainGraph
    from stencilflow.kernel import Kernel
    from stencilflow.input import Input
    from stencilflow.output import Output

    stencil_dir = os.path.join(REFERENCE, "test", "stencils")
    names = sorted(f[:-5] for f in os.listdir(stencil_dir)
                   if f.endswith(".json"))
    structure = {}
    for name in names:
        if name == "simple_input_delay_buf":
            continue  # stale fixture: assigns `res`, not the kernel name
        path = os.path.join(stencil_dir, name + ".json")
        try:
            chain = KernelChainGraph(path)
        except Exception as exc:  # record, do not hide
            structure[name] = {"error": type(exc).__name__ + ": " + str(exc)}
            continue
        import networkx as nx
        entry = {
            "dimensions": list(map(int, chain.dimensions)),
            "kernel_dimensions": int(chain.kernel_dimensions),
            "vectorization": int(chain.vectorization),
            "outputs": list(chain.outputs),
            "kernels": {},
            "inputs": {},
            "edges": sorted([type(u).__name__, u.name,
                             type(v).__name__, v.name]
                            for u, v in chain.graph.edges()),
            "topological_kernels": [
                n.name for n in nx.topological_sort(chain.graph)
                if isinstance(n, Kernel)
            ],
            "minimum_communication_volume":
            int(chain.minimum_communication_volume()),
            "operation_count": {
                k: [int(v[0]), int(v[1])]
                for k, v in chain.operation_count().items()
            },
        }
        for kname, k in chain.kernel_nodes.items():
            entry["kernels"][kname] = {
                "data_type": repr(k.data_type),
                "kernel_string": k.kernel_string,
                "accesses": {f: [list(ix) for ix in lst]
                             for f, lst in k.graph.accesses.items()},
                "reads": sorted(k.inputs.keys()),
                "input_dims": {
                    f: (list(ch["input_dims"])
                        if ch.get("input_dims") is not None else None)
                    for f, ch in k.inputs.items() if "input_dims" in ch
                },
            }
        for iname, node in chain.input_nodes.items():
            entry["inputs"][iname] = {
                "data_type": repr(node.data_type),
                "input_dims": list(chain.inputs[iname]["input_dims"]),
            }
        structure[name] = entry
    with open(os.path.join(HERE, "reference_structure.json"), "w") as f:
        json.dump(structure, f, indent=1, sort_keys=True)
    print("wrote reference_structure.json with", len(structure), "programs")

    # The one numeric vector the reference can produce here: its Simulator on
    # simulator12.json (3x3x3, W=1) -- SURVEY.md §8c.
    from stencilflow.simulator import Simulator
    from stencilflow.log_level import LogLevel
    path = os.path.join(stencil_dir, "simulator12.json")
    desc = stencilflow.parse_json(path)
    chain = KernelChainGraph(path)
    sim = Simulator(program_name="simulator12",
                    program_description=desc,
                    input_nodes=chain.input_nodes,
                    kernel_nodes=chain.kernel_nodes,
                    output_nodes=chain.output_nodes,
                    dimensions=chain.dimensions,
                    write_output=False,
                    log_level=LogLevel.NO_LOG)
    sim.simulate()
    result = {k: [float(x) for x in np.ravel(v)]
              for k, v in sim.get_result().items()}
    with open(os.path.join(HERE, "simulator12_expected.json"), "w") as f:
        json.dump({"program": "simulator12", "dimensions": [3, 3, 3],
                   "source": "reference stencilflow.simulator.Simulator",
                   "result": result}, f, indent=1)
    print("wrote simulator12_expected.json:", result)


if __name__ == "__main__":
    main()
