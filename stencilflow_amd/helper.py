"""Program-description I/O helpers shared by the operator API and the driver.

Mirrors the *behaviour* (names, argument meaning, file formats, exceptions) of
the reference's ``stencilflow/helper.py`` for the functions the stencil-chain
compute path touches:

* ``parse_json``            reference helper.py:62-92
* ``load_array``            reference helper.py:162-217
* ``load_input_arrays``     reference helper.py:220-237
* ``save_output_arrays``    reference helper.py:249-258
* ``arrays_are_equal``      reference helper.py:261-276
* ``aligned``               reference helper.py:328-338
* ``dim_to_abs_val`` / ``convert_3d_to_1d`` / ``num_dims``  helper.py:147-159,293-325
* ``OpCounter``             reference helper.py:341-365
"""

import ast
import collections
import functools
import json
import operator
import os
import re

import numpy as np

from .dtypes import str_to_dtype, typeclass

ITERATORS = ["i", "j", "k"]

_PACKAGE_DIR = os.path.dirname(os.path.realpath(__file__))


def parse_json(config_path):
    """Read a program (or config) file; ``data_type`` strings become type objects.

    Missing file -> ``RuntimeError`` (reference helper.py:69-74); unknown dtype
    -> ``AttributeError`` (helper.py:59).  The directory of the file is stored
    under ``"path"`` (helper.py:80).
    """
    located = next((c for c in (config_path, os.path.join(_PACKAGE_DIR, config_path))
                    if os.path.isfile(c)), None)
    if located is None:
        raise RuntimeError("file {} does not exists.".format(config_path))
    with open(located, "r") as handle:
        config = json.load(handle)
    config["path"] = os.path.dirname(os.path.abspath(located))
    _convert_dtypes(config)
    return config


def _convert_dtypes(tree):
    for key, val in tree.items():
        if isinstance(val, dict):
            _convert_dtypes(val)
        elif key == "data_type" and not isinstance(val, typeclass):
            tree[key] = str_to_dtype(val)


def _require(value, kind, label):
    if not isinstance(value, kind):
        raise Exception("{} should be of type {}, but is of type {}".format(label, kind, type(value)))


def max_dict_entry_key(dict1):
    """Key of the largest value."""
    _require(dict1, dict, "dict1")
    best = None
    for key, value in dict1.items():
        if best is None or value > dict1[best]:
            best = key
    if best is None:
        raise ValueError("max() arg is an empty sequence")
    return best


def list_add_cwise(list1, list2):
    _require(list1, list, "list1")
    _require(list2, list, "list2")
    return [x + y for x, y in zip(list1, list2)]


def list_subtract_cwise(list1, list2):
    """Element-wise difference; ``None`` wherever either side is ``None``."""
    _require(list1, list, "list1")
    _require(list2, list, "list2")
    return [None if (x is None or y is None) else x - y for x, y in zip(list1, list2)]


def dim_to_abs_val(input, dimensions):
    """Flatten ``[x, y, z]`` against C-order ``dimensions`` (helper.py:147-159)."""
    strides = [
        functools.reduce(operator.mul, dimensions[d + 1:], 1)
        for d in range(len(dimensions))
    ]
    return sum(a * s for a, s in zip(input, strides))


def num_dims(index):
    return sum(1 for x in index if x is not None)


def convert_3d_to_1d(dimensions, index):
    """Flat C-order offset of a 3-D index with ``None`` for absent dims."""
    if not index:
        return 0
    n = num_dims(index)
    if n == 3:
        return dim_to_abs_val(index, dimensions)
    if n == 2:
        if index[0] is None:
            return index[1] * dimensions[2] + index[2]
        if index[1] is None:
            return index[0] * dimensions[2] + index[2]
        return index[0] * dimensions[1] + index[1]
    if n == 1:
        return [x for x in index if x is not None][0]
    return 0


_GENERATED = re.compile(r"([^:]+):(.+)")


def _is_scalar_input(input_config):
    dims = input_config.get("input_dims")
    return dims is not None and len(dims) == 0


def _generate(kind, argument, dtype, shape, scalar):
    if kind == "constant":
        value = float(argument)
        return value if scalar else np.full(shape, value, dtype=dtype)
    if kind == "random":
        fields = [t for t in re.split(r"[,\s]+|\.\.", argument) if t]
        low, high = float(fields[0]), float(fields[1])
        rng = np.random.default_rng(int(fields[2]) if len(fields) > 2 else 0)
        return float(rng.uniform(low, high)) if scalar else rng.uniform(low, high, shape).astype(dtype)
    raise ValueError("Unknown generation: " + kind)


def _read_file(name, prefix, dtype):
    candidates = [name] + ([os.path.join(prefix, name)] if prefix is not None else [])
    path = next((c for c in candidates if os.path.isfile(c)), None)
    if path is None:
        raise FileNotFoundError("File {} does not exists.".format(name))
    extension = os.path.splitext(path)[1]
    if extension == ".csv":
        return np.genfromtxt(path, dtype, delimiter=",")
    if extension == ".dat":
        return np.fromfile(path, dtype)
    raise ValueError("Invalid file type: " + path)


def load_array(input_config, prefix=None, shape=None):
    """Materialise one program input (reference helper.py:162-217).

    ``data`` may be ``"constant:<v>"``, ``"random:<lo>,<hi>[,<seed>]"`` (the
    reference's ``random:`` branch raises ``NameError``; a working one is
    provided), a ``.csv``/``.dat`` path (looked up under ``prefix`` as well), an
    inline list, or a plain number for 0-D inputs.
    """
    data = input_config["data"]
    dtype = input_config["data_type"].type
    scalar = _is_scalar_input(input_config)
    if isinstance(data, str):
        generated = None if os.path.isfile(data) else _GENERATED.match(data)
        if generated is None:
            return _read_file(data, prefix, dtype)
        if shape is None and not scalar:
            raise ValueError("Must provide shape when using generated inputs")
        return _generate(generated.group(1), generated.group(2), dtype, shape, scalar)
    if scalar or (shape is not None and len(shape) == 0):
        return dtype(data)
    return data if isinstance(data, np.ndarray) else np.array(data, dtype=dtype)


def load_input_arrays(input_configs, prefix=None, shape=None):
    """All program inputs, arrays 64-byte aligned (helper.py:220-237)."""
    loaded = {}
    for name, source in input_configs.items():
        value = load_array(source, prefix, shape)
        is_array = isinstance(value, np.ndarray) and value.ndim > 0
        loaded[name] = aligned(value, 64) if is_array else value
    return loaded


def save_array(array, path):
    """Raw C-order dump."""
    np.ascontiguousarray(array).tofile(path)


def save_output_arrays(outputs, output_dir=str()):
    """``<name>.dat`` raw C-order dump per output (helper.py:249-258)."""
    for name, data in outputs.items():
        save_array(data, os.path.join(output_dir, name + ".dat"))


def arrays_are_equal(reference, result, tolerance=1e-5):
    """The reference's comparison rule, bug-compatible (helper.py:261-276).

    ``|ref - res| / (max(ref, res) + eps) <= tol`` with a *signed* denominator,
    so negative data passes trivially.  ``arrays_match`` below is the strict
    rule this backend's own parity tests use.
    """
    ref, res = (a if isinstance(a, np.ndarray) else load_array(a) for a in (reference, result))
    denominator = np.maximum(ref, res) + np.finfo(ref.dtype).eps  # signed, as in the reference
    return np.all(np.abs(ref - res) / denominator <= tolerance)


def arrays_match(reference, result, tolerance=1e-6):
    """Strict relative comparison: ``|ref-res| <= tol * max(|ref|,|res|)``,
    NaNs never match, exact zeros must match to ``tol * tiny``."""
    reference = np.asarray(reference)
    result = np.asarray(result)
    if reference.shape != result.shape:
        return False
    if np.isnan(reference).any() or np.isnan(result).any():
        return False
    scale = np.maximum(np.abs(reference), np.abs(result))
    return bool(np.all(np.abs(reference - result) <= tolerance * scale))


def unique(iterable):
    """Distinct elements in first-occurrence order, in a container of the same
    type; unhashable elements are told apart by their ``str``."""
    seen, kept = set(), []
    for item in iterable:
        try:
            key = item
            hash(key)
        except TypeError:
            key = str(item)
        if key not in seen:
            seen.add(key)
            kept.append(item)
    return type(iterable)(kept)


def aligned(a, alignment=16):
    """Return ``a`` or a copy whose base address is ``alignment``-aligned."""
    if a.ctypes.data % alignment:
        spare = alignment // a.itemsize + 1
        pool = np.empty(a.size + spare, dtype=a.dtype)
        first = (-pool.ctypes.data % alignment) // a.itemsize
        copy = pool[first:first + a.size].reshape(a.shape)
        copy[...] = a
        if copy.ctypes.data % alignment:
            raise MemoryError("could not align a {}-byte element array to {}".format(a.itemsize, alignment))
        return copy
    return a


class OpCounter(ast.NodeVisitor):
    """Counts arithmetic operations the way reference helper.py:341-365 does:
    a ``BinOp`` is counted when at least one operand is a field access or
    another ``BinOp``; every ``Call`` counts under its function name."""

    def __init__(self):
        self._counts = collections.Counter()

    @property
    def operation_count(self):
        return dict(self._counts)

    def visit_BinOp(self, node):
        operands = (node.left, node.right)
        if any(isinstance(x, (ast.Subscript, ast.BinOp)) for x in operands):
            self._counts[type(node.op).__name__] += 1
        for x in operands:
            self.visit(x)

    def visit_Call(self, node):
        self._counts[node.func.id] += 1
        for argument in node.args:
            self.visit(argument)
