"""Operator API of the stencil-chain path: the program DAG.

``KernelChainGraph`` keeps the public surface of the reference class
(stencilflow/kernel_chain_graph.py:32-114): same constructor, same attribute
names with the same meaning (``graph``, ``kernel_nodes``, ``input_nodes``,
``output_nodes``, ``dimensions``, ``kernel_dimensions``, ``vectorization``,
``constants``, ``inputs``, ``outputs``, ``program``), the same node classes
(``Input`` / ``Kernel`` / ``Output``) in a ``networkx.DiGraph`` and the same edge
rule (an edge wherever a consumer reads a field named like the producer,
kernel_chain_graph.py:243-272).  What is *not* carried over is the FPGA
buffer model behind it (delay buffers, channel depths, latencies, simulator
state): a GPU backend has no use for it.

Edges are matched through a name table, so construction is linear in the
number of stages (the reference's double loop is quadratic: 41 s for a
1000-stage chain).
"""

import ast
import copy
import functools
import operator
import os

import networkx as nx

from . import dtypes
from .expr import KernelExpr, to_c
from .helper import ITERATORS, OpCounter, parse_json
from .log_level import LogLevel


class _Node:
    """Fields every DAG node carries (reference base_node_class.py:53-83)."""

    def __init__(self, name, data_type):
        if not isinstance(data_type, dtypes.typeclass):
            raise TypeError("Expected typeclass, got: " +
                            type(data_type).__name__)
        self.name = name
        self.data_type = data_type
        self.inputs = dict()  # predecessor name -> channel dict
        self.outputs = dict()  # successor name -> channel dict

    def generate_label(self):
        return self.name

    def __repr__(self):
        return "{}({})".format(type(self).__name__, self.name)


class Input(_Node):
    """A program input array or scalar (reference stencilflow/input.py)."""


class Output(_Node):
    """A program output (reference stencilflow/output.py)."""

    def __init__(self, name, data_type, dimensions):
        super().__init__(name, data_type)
        self.dimensions = dimensions


class _AccessView:
    """The slice of the reference's ``ComputeGraph`` that lowering reads:
    ``accesses`` (compute_graph.py:125-144), ``min_index`` / ``max_index``."""

    def __init__(self, expr):
        self.accesses = {
            f: [list(ix) for ix in lst]
            for f, lst in expr.accesses.items()
        }
        for s in expr.scalars:
            self.accesses[s] = [[0, 0, 0]]
        self.min_index = {}
        self.max_index = {}
        for f, lst in self.accesses.items():
            comparable = [[-10**9 if v is None else v for v in ix]
                          for ix in lst]
            self.min_index[f] = lst[comparable.index(min(comparable))]
            self.max_index[f] = lst[comparable.index(max(comparable))]


class Kernel(_Node):
    """One stencil operator (reference stencilflow/kernel.py:27-78).

    ``kernel_string`` is the text the reference would hold (with the ``i``
    iterator spliced into 1-D/2-D accesses, kernel_chain_graph.py:392-405);
    ``expr`` is this backend's typed tree of the same statements.
    """

    def __init__(self, name, kernel_string, computation_string, dimensions,
                 data_type, boundary_conditions, raw_inputs, field_info,
                 scalar_info, vectorization=1):
        super().__init__(name, data_type)
        self.kernel_string = kernel_string
        self.computation_string = computation_string
        self.dimensions = dimensions
        self.boundary_conditions = boundary_conditions
        self.raw_inputs = raw_inputs
        self.vectorization = vectorization
        self.expr = KernelExpr(name, computation_string, field_info,
                               scalar_info, boundary_conditions)
        self.graph = _AccessView(self.expr)

    def generate_relative_access_kernel_string(self,
                                               relative_to_center=True,
                                               replace_negative_index=False,
                                               python_syntax=False,
                                               flatten_index=True,
                                               output_dimensions=None):
        """Re-emitted, fully parenthesised statements with relative indices
        (role of reference kernel.py:325-368; C syntax only)."""
        rename = {}
        for f, lst in self.expr.accesses.items():
            for ix in lst:
                rename[(f, ix)] = "{}[{}]".format(
                    f, ", ".join(str(o) for o in ix if o is not None))
        return "; ".join("{} = {}".format(t, to_c(n, rename))
                         for t, n in self.expr.statements)


class KernelChainGraph:
    def __init__(self, path, plot_graph=False, log_level=LogLevel.NO_LOG):
        if log_level >= LogLevel.MODERATE:
            print("Initialize KernelChainGraph.")
        self.path = os.path.abspath(path)
        self.log_level = log_level
        self.inputs = dict()
        self.outputs = list()
        self.dimensions = list()
        self.program = dict()
        self.vectorization = 1
        self.graph = nx.DiGraph()
        self.input_nodes = dict()
        self.output_nodes = dict()
        self.kernel_nodes = dict()
        self.channels = dict()
        self.name = os.path.splitext(os.path.basename(self.path))[0]
        self.kernel_dimensions = -1
        self.constants = {}
        self.import_input()
        self.create_kernels()
        self.connect_kernels()
        self.check_acyclic()
        if plot_graph:
            self.plot_graph(self.name + ".png")
        if self.log_level >= LogLevel.MODERATE:
            self.report(self.name)

    # ------------------------------------------------------------------ parse
    def import_input(self):
        """Program file -> fields (reference kernel_chain_graph.py:364-407):
        dimensions padded to 3-D, default ``input_dims`` = the last
        ``kernel_dimensions`` iterators, ``i`` spliced into 1-D/2-D accesses."""
        inp = parse_json(self.path)
        self.kernel_dimensions = len(inp["dimensions"])
        if not 1 <= self.kernel_dimensions <= 3:
            raise ValueError("Programs must have 1 to 3 dimensions")
        self.constants = copy.copy(inp["constants"]) if "constants" in inp \
            else {}
        self.vectorization = int(
            inp["vectorization"]) if "vectorization" in inp else 1
        self.program = inp["program"]
        self.inputs = inp["inputs"]
        own_iterators = ITERATORS[len(ITERATORS) - self.kernel_dimensions:]
        for desc in self.inputs.values():
            if "input_dims" not in desc:
                desc["input_dims"] = list(desc["dimensions"]) \
                    if "dimensions" in desc else list(own_iterators)
        self.outputs = inp["outputs"]
        self._original_strings = {
            k: str(v["computation_string"])
            for k, v in self.program.items()
        }
        pad = len(ITERATORS) - self.kernel_dimensions
        if pad:
            splice = "[i," if pad == 1 else "[i, j,"
            for entry in self.program.values():
                entry["computation_string"] = entry[
                    "computation_string"].replace("[", splice)
        self.dimensions = [1] * pad + list(inp["dimensions"])

    def total_elements(self):
        return functools.reduce(operator.mul, self.dimensions, 1)

    def create_kernels(self):
        """One node per program entry / input / output
        (reference kernel_chain_graph.py:417-455)."""
        own_iterators = ITERATORS[len(ITERATORS) - self.kernel_dimensions:]
        field_info = {}
        scalar_info = {}
        for name, desc in self.inputs.items():
            dims = list(desc["input_dims"]) if desc["input_dims"] is not None \
                else list(own_iterators)
            if len(dims) == 0:
                scalar_info[name] = (desc["data_type"], "scalar")
            else:
                field_info[name] = (dims, desc["data_type"])
        for name, desc in self.constants.items():
            scalar_info[name] = (desc["data_type"], "constant")
        for name, desc in self.program.items():
            if name in field_info or name in scalar_info:
                raise ValueError(
                    "'{}' names both an input and a kernel".format(name))
            field_info[name] = (list(own_iterators), desc["data_type"])
        self.field_info = field_info
        self.scalar_info = scalar_info

        self.kernel_nodes = dict()
        for name, desc in self.program.items():
            node = Kernel(name=name,
                          kernel_string=str(desc["computation_string"]),
                          computation_string=self._original_strings[name],
                          dimensions=self.dimensions,
                          data_type=desc["data_type"],
                          boundary_conditions=desc["boundary_conditions"],
                          raw_inputs=self.inputs,
                          field_info=field_info,
                          scalar_info=scalar_info,
                          vectorization=self.vectorization)
            self.graph.add_node(node)
            self.kernel_nodes[name] = node
        self.input_nodes = dict()
        for name, desc in self.inputs.items():
            node = Input(name=name, data_type=desc["data_type"])
            self.input_nodes[name] = node
            self.graph.add_node(node)
        self.output_nodes = dict()
        for name in self.outputs:
            if name not in self.program:
                raise RuntimeError(
                    "Output '{}' is not produced by any kernel".format(name))
            node = Output(name=name,
                          data_type=self.program[name]["data_type"],
                          dimensions=self.dimensions)
            self.output_nodes[name] = node
            self.graph.add_node(node)

    def connect_kernels(self):
        """Edges by name (reference kernel_chain_graph.py:243-362): producer ->
        consumer for every field or scalar the consumer reads; kernel -> output
        of the same name.  Each edge carries a ``channel`` dict with ``name``,
        ``data_type`` and, for program inputs, ``input_dims``."""
        self.channels = dict()
        for dest in self.kernel_nodes.values():
            for read in dest.graph.accesses:
                if read in self.kernel_nodes:
                    src = self.kernel_nodes[read]
                    if src is dest:
                        raise ValueError(
                            "Cycle detected: {}".format([dest.name]))
                elif read in self.input_nodes:
                    src = self.input_nodes[read]
                else:
                    continue  # program constant
                channel = {
                    "name": src.name + "_" + dest.name,
                    "data_type": src.data_type
                }
                if isinstance(src, Input):
                    channel["input_dims"] = self.inputs[src.name].get(
                        "input_dims")
                self._add_edge(src, dest, channel)
        for name, dest in self.output_nodes.items():
            src = self.kernel_nodes[name]
            self._add_edge(src, dest, {
                "name": src.name + "_" + dest.name,
                "data_type": src.data_type
            })

    def _add_edge(self, src, dest, channel):
        self.channels[channel["name"]] = channel
        src.outputs[dest.name] = channel
        dest.inputs[src.name] = channel
        self.graph.add_edge(src, dest, channel=channel)

    def check_acyclic(self):
        """Same diagnostic as reference kernel_chain_graph.py:488-493."""
        try:
            self._order = list(nx.topological_sort(self.graph))
        except nx.exception.NetworkXUnfeasible:
            cycle = next(nx.algorithms.cycles.simple_cycles(self.graph))
            raise ValueError("Cycle detected: {}".format(
                [c.name for c in cycle]))

    def topological_kernels(self):
        """Kernels in the order the reference's CPU program runs them
        (``nx.topological_sort``, sdfg_generator.py:638-641)."""
        return [n for n in nx.topological_sort(self.graph)
                if isinstance(n, Kernel)]

    # --------------------------------------------------------------- analysis
    def operation_count(self):
        """op name -> (per point, total) as reference :721-747."""
        num_iterations = self.total_elements()
        operations = {}
        for kernel in self.graph.nodes():
            if not isinstance(kernel, Kernel):
                continue
            counter = OpCounter()
            counter.visit(ast.parse(kernel.kernel_string))
            for name, count in counter.operation_count.items():
                prev = operations.get(name, (0, 0))
                operations[name] = (prev[0] + count,
                                    prev[1] + num_iterations * count)
        return operations

    def minimum_communication_volume(self):
        """Bytes that must cross the off-chip interface at least once: every
        input once plus every output once (reference :749-768)."""
        volume = 0
        for v in self.inputs.values():
            elements = 1
            for it in v["input_dims"]:
                elements *= self.dimensions[ITERATORS.index(it)]
            volume += v["data_type"].bytes * elements
        for name in self.outputs:
            volume += self.program[name]["data_type"].bytes * \
                self.total_elements()
        return volume

    def cell_updates(self):
        """Operators x grid points: the unit of work of the GPU backend."""
        return self.total_elements() * len(self.kernel_nodes)

    def algorithmic_bytes(self):
        """``2 * sizeof(dtype)`` per cell update."""
        return sum(2 * k.data_type.bytes * self.total_elements()
                   for k in self.kernel_nodes.values())

    def runtime_lower_bound(self, hbm_bytes_per_s=8.0e12):
        """Seconds the chain needs at the HBM roofline if every operator made
        one read and one write pass (the GPU analogue of reference :770-774,
        which counts FPGA cycles)."""
        return self.algorithmic_bytes() / hbm_bytes_per_s

    def report(self, name=None):
        print("Report of {}".format(name or self.name))
        print("  dimensions: {}  ({} points)".format(self.dimensions,
                                                     self.total_elements()))
        print("  kernels: {}  inputs: {}  outputs: {}".format(
            len(self.kernel_nodes), len(self.input_nodes),
            len(self.output_nodes)))
        for op, (per_point, total) in sorted(self.operation_count().items()):
            print("  {:>6}: {} per point, {} total".format(
                op, per_point, total))
        print("  minimum off-chip volume: {} bytes".format(
            self.minimum_communication_volume()))
        print("  algorithmic traffic (2*sizeof per update): {} bytes".format(
            self.algorithmic_bytes()))
        print("  runtime lower bound at 8 TB/s: {:.6f} s".format(
            self.runtime_lower_bound()))

    def plot_graph(self, save_path=None):
        raise NotImplementedError(
            "plotting is outside the compute path this backend provides")
