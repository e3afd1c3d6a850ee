"""Scalar type objects for the stencil program description.

The reference stores ``dace.dtypes.typeclass`` objects wherever a program
names a ``data_type`` (reference: stencilflow/helper.py:47-59 ``str_to_dtype``;
consumers read ``.type`` (numpy scalar type), ``.bytes`` and call the object to
cast a Python number, e.g. stencilflow/run_program.py:153-158,
stencilflow/kernel_chain_graph.py:755-766).  DaCe is not a dependency of this
backend, so the same small surface is provided here.
"""

import numpy as np


class typeclass:
    """Look-alike of the part of ``dace.dtypes.typeclass`` StencilFlow uses."""

    __slots__ = ("name", "type", "bytes", "ctype", "rank", "is_float")

    def __init__(self, name, nptype, ctype, rank, is_float):
        self.name = name
        self.type = nptype
        self.bytes = np.dtype(nptype).itemsize
        self.ctype = ctype
        # rank orders the types for C++ "usual arithmetic conversions"
        self.rank = rank
        self.is_float = is_float

    def __call__(self, value):
        return self.type(value)

    def __repr__(self):
        return self.name

    def __eq__(self, other):
        return isinstance(other, typeclass) and other.name == self.name

    def __hash__(self):
        return hash(self.name)

    def to_string(self):
        return self.name

    def as_numpy_dtype(self):
        return np.dtype(self.type)


bool_ = typeclass("bool", np.bool_, "bool", 0, False)
int8 = typeclass("int8", np.int8, "signed char", 1, False)
uint8 = typeclass("uint8", np.uint8, "unsigned char", 1, False)
int16 = typeclass("int16", np.int16, "short", 2, False)
uint16 = typeclass("uint16", np.uint16, "unsigned short", 2, False)
int32 = typeclass("int32", np.int32, "int", 3, False)
uint32 = typeclass("uint32", np.uint32, "unsigned int", 4, False)
int64 = typeclass("int64", np.int64, "long long", 5, False)
uint64 = typeclass("uint64", np.uint64, "unsigned long long", 6, False)
float32 = typeclass("float32", np.float32, "float", 10, True)
float64 = typeclass("float64", np.float64, "double", 11, True)

_ALL = {
    t.name: t
    for t in (int8, uint8, int16, uint16, int32, uint32, int64, uint64,
              float32, float64)
}
_ALL["bool"] = bool_


def str_to_dtype(dtype_str):
    """Name -> type object; error behaviour of reference helper.py:47-59."""
    if not isinstance(dtype_str, str):
        raise TypeError("Expected string, got: " + type(dtype_str).__name__)
    try:
        return _ALL[dtype_str]
    except KeyError:
        pass
    raise AttributeError("Unsupported data type: " + dtype_str)


def promote(a, b):
    """C++ usual arithmetic conversions for a binary operator on (a, b).

    Integer types narrower than ``int`` (and bool) promote to ``int`` first.
    This is the typing the C++ emitted by DaCe for a Python tasklet obeys
    (reference: stencilflow/stencil/cpu.py:46-115 builds the tasklet; DaCe
    lowers it to C++), and therefore the typing contract of this backend.
    """
    if a.is_float or b.is_float:
        if a.is_float and b.is_float:
            return a if a.rank >= b.rank else b
        return a if a.is_float else b
    a = int32 if a.rank < int32.rank else a
    b = int32 if b.rank < int32.rank else b
    return a if a.rank >= b.rank else b


def literal_type(value):
    """Type of a Python literal once printed into C++ source."""
    if isinstance(value, bool):
        return int32
    if isinstance(value, int):
        return int32 if -2**31 <= value < 2**31 else int64
    if isinstance(value, float):
        return float64
    raise TypeError("Unsupported literal: {!r}".format(value))
