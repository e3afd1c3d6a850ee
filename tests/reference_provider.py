"""Registers the oracle as the CPU checker of ``run_program``'s
``compare_to_reference`` (test infrastructure; the product never imports
``oracle/``)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def reference_outputs(stencil_file, input_arrays):
    from oracle import numpy_oracle
    return numpy_oracle.run_reference(stencil_file, inputs=input_arrays)


def register():
    from stencilflow_amd.run_program import set_reference_backend
    set_reference_backend(reference_outputs)
