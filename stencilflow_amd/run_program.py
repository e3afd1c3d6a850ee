"""Driver with the reference's ``run_program`` contract
(stencilflow/run_program.py:19-250), executing on an MI355X through
``libsf_hip.so`` instead of an FPGA through DaCe.

Same signature, same return values (``0`` verified, ``None`` when execution is
skipped), same files written (``results/<name>/<out>.dat`` and
``results/<name>/reference/<out>.dat``), same ``ValueError("Result
mismatch.")``.  Flags that only configure the FPGA toolchain
(``use_cached_sdfg``, ``synthetic_reads``, ``specialize_scalars``, ``xilinx``)
are accepted and have no effect; both ``mode`` values run the HIP backend
(there is no emulator: ``"emulation"`` is kept so existing callers work, and
``"hip"`` is accepted as an alias of ``"hardware"``).

The CPU reference that ``compare_to_reference`` checks against is *not* part of
this package: a checker must be registered with ``set_reference_backend`` (the
test-suite registers the oracle under ``oracle/``).  Without one the flag
raises ``RuntimeError`` — the product path never computes on the CPU.
"""

import copy
import itertools
import os
import re

import numpy as np

from . import helper
from .backend import compile_program
from .kernel_chain_graph import KernelChainGraph
from .log_level import LogLevel

_REFERENCE_BACKEND = None


def set_reference_backend(fn):
    """Register the CPU checker: ``fn(stencil_file, input_arrays) -> dict`` of
    output name -> ndarray.  Returns the previous one."""
    global _REFERENCE_BACKEND
    prev = _REFERENCE_BACKEND
    _REFERENCE_BACKEND = fn
    return prev


def run_program(stencil_file,
                mode,
                run_simulation=False,
                compare_to_reference=False,
                input_directory=None,
                use_cached_sdfg=None,
                skip_execution=False,
                generate_input=False,
                synthetic_reads=None,
                specialize_scalars=False,
                plot=False,
                halo=0,
                repetitions=1,
                log_level=LogLevel.BASIC,
                print_result=False,
                xilinx=False,
                device=0,
                options=None,
                tolerance=1e-6):
    def log(msg, level=LogLevel.BASIC):
        if log_level >= level:
            print(msg)

    program_description = helper.parse_json(stencil_file)
    name = os.path.basename(stencil_file)
    name = re.match(r"(.+)\.[^\.]+", name).group(1).replace(".", "_")

    log("Creating kernel graph...")
    chain = KernelChainGraph(path=stencil_file,
                             plot_graph=plot,
                             log_level=log_level)

    if run_simulation:
        raise NotImplementedError(
            "The cycle-level FPGA simulator is not part of the HIP backend")
    if mode not in ("emulation", "hardware", "hip"):
        raise ValueError("Unrecognized execution mode: {}".format(mode))
    if compare_to_reference and _REFERENCE_BACKEND is None:
        raise RuntimeError(
            "compare_to_reference needs a CPU checker: register one with "
            "stencilflow_amd.run_program.set_reference_backend()")

    log("Generating and compiling HIP kernels...")
    program = compile_program(chain, device=device, options=options)
    log(program.plan.describe(), LogLevel.MODERATE)

    if skip_execution or repetitions == 0:
        log("Skipping execution and exiting.")
        program.close()
        return

    log("Loading input arrays...")
    if input_directory is None:
        input_directory = os.path.dirname(stencil_file)
    input_description = copy.copy(program_description["inputs"])
    if generate_input:
        # reference run_program.py:141-144
        for k in input_description:
            input_description[k] = dict(input_description[k])
            input_description[k]["data"] = "constant:0.5"
    input_arrays = {}
    for arr_name, source in input_description.items():
        source = dict(source)
        source["input_dims"] = chain.inputs[arr_name]["input_dims"]
        dims = source["input_dims"]
        own = helper.ITERATORS[3 - chain.kernel_dimensions:]
        shape = [program_description["dimensions"][own.index(d)] for d in dims]
        arr = helper.load_array(source, prefix=input_directory, shape=shape)
        if isinstance(arr, np.ndarray) and arr.ndim > 0:
            arr = helper.aligned(
                np.ascontiguousarray(arr.reshape(shape)), 64)
        input_arrays[arr_name] = arr

    log("Initializing output arrays...")
    output_arrays = {
        arr_name: helper.aligned(
            np.zeros(program_description["dimensions"],
                     dtype=program_description["program"][arr_name]
                     ["data_type"].type), 64)
        for arr_name in program_description["outputs"]
    }

    # arrays are keyed "<name>_host", 0-D inputs by bare name (reference :164-169)
    args = {(key + "_host" if hasattr(val, "shape") and len(val.shape) > 0
             else key): val
            for key, val in itertools.chain(input_arrays.items(),
                                            output_arrays.items())}
    if repetitions == 1:
        log("Executing program on the GPU...")
        program(**args)
        log("Finished running program.")
    else:
        for i in range(repetitions):
            log("Executing repetition {}/{}...".format(i + 1, repetitions))
            program(**args)
            log("Finished running program.")
    program.close()

    if print_result:
        for key, val in output_arrays.items():
            print(key + ":", val)

    reference_output_arrays = None
    if compare_to_reference:
        log("Executing reference program...")
        reference_output_arrays = _REFERENCE_BACKEND(stencil_file,
                                                     input_arrays)
        log("Finished running program.")
        if print_result:
            for key, val in reference_output_arrays.items():
                print(key + ":", val)

    output_folder = os.path.join("results", name)
    os.makedirs(output_folder, exist_ok=True)
    if halo > 0:
        # prune halos (reference :202-209)
        for k, v in output_arrays.items():
            output_arrays[k] = v[tuple(slice(halo, -halo) for _ in v.shape)]
        if compare_to_reference:
            for k, v in reference_output_arrays.items():
                reference_output_arrays[k] = v[tuple(
                    slice(halo, -halo) for _ in v.shape)]
    helper.save_output_arrays(output_arrays, output_folder)
    log("Results saved to " + output_folder)
    if compare_to_reference:
        reference_folder = os.path.join(output_folder, "reference")
        os.makedirs(reference_folder, exist_ok=True)
        helper.save_output_arrays(reference_output_arrays, reference_folder)
        log("Reference results saved to " + reference_folder)

    if compare_to_reference:
        log("Comparing to reference...")
        for outp in output_arrays:
            got = output_arrays[outp]
            expected = reference_output_arrays[outp]
            if not helper.arrays_match(np.ravel(expected), np.ravel(got),
                                       tolerance):
                print("Expected: {}".format(expected))
                print("Got:      {}".format(got))
                raise ValueError("Result mismatch.")
        log("Results verified.")
        return 0
    return None
