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
    if not os.path.isfile(config_path):
        candidate = os.path.join(_PACKAGE_DIR, config_path)
        if not os.path.isfile(candidate):
            raise RuntimeError("file {} does not exists.".format(config_path))
        config_path = candidate
    with open(config_path, "r") as handle:
        config = json.load(handle)
    config["path"] = os.path.dirname(os.path.abspath(config_path))
    _convert_dtypes(config)
    return config


def _convert_dtypes(tree):
    for key, val in tree.items():
        if isinstance(val, dict):
            _convert_dtypes(val)
        elif key == "data_type" and not isinstance(val, typeclass):
            tree[key] = str_to_dtype(val)


def max_dict_entry_key(dict1):
    if not isinstance(dict1, dict):
        raise Exception("dict1 should be of type {}, but is of type {}".format(
            dict, type(dict1)))
    return max(dict1, key=dict1.get)


def list_add_cwise(list1, list2):
    for name, lst in (("list1", list1), ("list2", list2)):
        if not isinstance(lst, list):
            raise Exception("{} should be of type {}, but is of type {}".format(
                name, list, type(lst)))
    return [x + y for x, y in zip(list1, list2)]


def list_subtract_cwise(list1, list2):
    for name, lst in (("list1", list1), ("list2", list2)):
        if not isinstance(lst, list):
            raise Exception("{} should be of type {}, but is of type {}".format(
                name, list, type(lst)))
    return [
        x - y if x is not None and y is not None else None
        for x, y in zip(list1, list2)
    ]


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


def load_array(input_config, prefix=None, shape=None):
    """Materialise one program input (reference helper.py:162-217).

    ``data`` may be ``"constant:<v>"``, ``"random:<lo>,<hi>[,<seed>]"`` (the
    reference's ``random:`` branch raises ``NameError``; a working one is
    provided), a ``.csv``/``.dat`` path (looked up under ``prefix`` as well), an
    inline list, or a plain number for 0-D inputs.
    """
    data = input_config["data"]
    dtype = input_config["data_type"].type
    is_scalar = ("input_dims" in input_config
                 and input_config["input_dims"] is not None
                 and len(input_config["input_dims"]) <= 0)
    if isinstance(data, str):
        m = _GENERATED.match(data)
        if m and not os.path.isfile(data):
            if shape is None and not is_scalar:
                raise ValueError(
                    "Must provide shape when using generated inputs")
            kind, arg = m.group(1), m.group(2)
            if kind == "constant":
                val = float(arg)
                if is_scalar:
                    return val
                arr = np.empty(shape, dtype=dtype)
                arr[:] = val
                return arr
            if kind == "random":
                parts = [p for p in re.split(r"[,\s]+|\.\.", arg) if p]
                lo, hi = float(parts[0]), float(parts[1])
                seed = int(parts[2]) if len(parts) > 2 else 0
                rng = np.random.default_rng(seed)
                if is_scalar:
                    return float(rng.uniform(lo, hi))
                return rng.uniform(lo, hi, shape).astype(dtype)
            raise ValueError("Unknown generation: " + kind)
        path = data
        if not os.path.isfile(path):
            if prefix is not None:
                path = os.path.join(prefix, path)
            if not os.path.isfile(path):
                raise FileNotFoundError("File {} does not exists.".format(data))
        if path.endswith(".csv"):
            return np.genfromtxt(path, dtype, delimiter=",")
        if path.endswith(".dat"):
            return np.fromfile(path, dtype)
        raise ValueError("Invalid file type: " + path)
    if is_scalar or (shape is not None and len(shape) == 0):
        return dtype(data)
    if isinstance(data, np.ndarray):
        return data
    return np.array(data, dtype=dtype)


def load_input_arrays(input_configs, prefix=None, shape=None):
    """All program inputs, arrays 64-byte aligned (helper.py:220-237)."""
    arrays = dict()
    for name, source in input_configs.items():
        arr = load_array(source, prefix, shape)
        if isinstance(arr, np.ndarray) and arr.ndim > 0:
            arr = aligned(arr, 64)
        arrays[name] = arr
    return arrays


def save_array(array, path):
    array.tofile(path)


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
    if not isinstance(reference, np.ndarray):
        reference = load_array(reference)
    if not isinstance(result, np.ndarray):
        result = load_array(result)
    relative_diff = (np.abs(reference - result) /
                     (np.maximum.reduce([reference, result]) +
                      np.finfo(reference.dtype).eps))
    return np.all(relative_diff <= tolerance)


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
    try:
        return type(iterable)(
            [i for i in sorted(set(iterable), key=lambda x: iterable.index(x))])
    except TypeError:
        return type(iterable)(collections.OrderedDict(
            zip(map(str, iterable), iterable)).values())


def aligned(a, alignment=16):
    """Return ``a`` or a copy whose base address is ``alignment``-aligned."""
    if (a.ctypes.data % alignment) == 0:
        return a
    extra = alignment // a.itemsize + 1
    buf = np.empty(a.size + extra, dtype=a.dtype)
    ofs = (-buf.ctypes.data % alignment) // a.itemsize
    view = buf[ofs:ofs + a.size].reshape(a.shape)
    np.copyto(view, a)
    assert view.ctypes.data % alignment == 0
    return view


class OpCounter(ast.NodeVisitor):
    """Counts arithmetic operations the way reference helper.py:341-365 does:
    a ``BinOp`` is counted when at least one operand is a field access or
    another ``BinOp``; every ``Call`` counts under its function name."""

    def __init__(self):
        self._operation_count = {}

    @property
    def operation_count(self):
        return self._operation_count

    def _bump(self, name):
        self._operation_count[name] = self._operation_count.get(name, 0) + 1

    def visit_BinOp(self, node):
        if any(
                isinstance(side, (ast.Subscript, ast.BinOp))
                for side in (node.left, node.right)):
            self._bump(type(node.op).__name__)
        self.generic_visit(node)

    def visit_Call(self, node):
        self._bump(node.func.id)
        self.generic_visit(node)
