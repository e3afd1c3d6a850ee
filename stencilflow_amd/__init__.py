"""MI355X-native execution backend for StencilFlow stencil-chain programs.

Drop-in for the compute path of the reference's ``run_program``
(stencilflow/run_program.py:64-178): same program JSON, same
``KernelChainGraph`` operator API, same driver signature; the operators run as
HIP kernels for gfx950 behind the C-ABI declared in ``include/sf_hip.h``.
"""

from .helper import *  # noqa: F401,F403
from .helper import (ITERATORS, aligned, arrays_are_equal, arrays_match,
                     load_array, load_input_arrays, parse_json,
                     save_output_arrays)
from .dtypes import str_to_dtype
from .log_level import LogLevel
from .kernel_chain_graph import Input, Kernel, KernelChainGraph, Output
from .run_program import run_program, set_reference_backend
from .run_distributed_program import run_distributed_program
