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

import os

import numpy as np

from . import helper
from .backend import compile_program
from .kernel_chain_graph import KernelChainGraph
from .log_level import LogLevel

_REFERENCE_BACKEND = None
_MODES = ("emulation", "hardware", "hip")


def set_reference_backend(fn):
    """Register the CPU checker: ``fn(stencil_file, input_arrays) -> dict`` of
    output name -> ndarray.  Returns the previous one."""
    global _REFERENCE_BACKEND
    prev = _REFERENCE_BACKEND
    _REFERENCE_BACKEND = fn
    return prev


def load_reference_backend(spec):
    """Register the checker named ``module:function`` (the command line's
    ``-reference-checker``).  Opt-in only: nothing is looked up by default."""
    import importlib
    module, _, attr = spec.partition(":")
    if not module or not attr:
        raise ValueError("reference checker must be given as module:function, got {!r}".format(spec))
    fn = getattr(importlib.import_module(module), attr)
    if not callable(fn):
        raise ValueError("{} is not callable".format(spec))
    return set_reference_backend(fn)


class _Log:
    def __init__(self, level):
        self.level = level

    def __call__(self, text, level=LogLevel.BASIC):
        if self.level >= level:
            print(text)


def _program_name(stencil_file):
    """`results/` sub-directory of a program: the file name without its last
    extension, remaining dots replaced (reference run_program.py:66-67)."""
    stem = os.path.basename(stencil_file)
    if "." in stem:
        stem = stem[:stem.rindex(".")]
    return stem.replace(".", "_")


def _materialise_inputs(chain, description, directory, generate_input):
    """Host arrays (64-byte aligned, C order) for array inputs, Python scalars
    for 0-D inputs.  `generate_input` replaces every source by `constant:0.5`
    (reference run_program.py:141-144)."""
    own_iterators = helper.ITERATORS[3 - chain.kernel_dimensions:]
    extents = dict(zip(own_iterators, description["dimensions"]))
    arrays = {}
    for name, declared in description["inputs"].items():
        source = dict(declared)
        if generate_input:
            source["data"] = "constant:0.5"
        source["input_dims"] = chain.inputs[name]["input_dims"]
        shape = [extents[d] for d in source["input_dims"]]
        value = helper.load_array(source, prefix=directory, shape=shape)
        if isinstance(value, np.ndarray) and value.ndim > 0:
            value = helper.aligned(np.ascontiguousarray(value.reshape(shape)), 64)
        arrays[name] = value
    return arrays


def _zeroed_outputs(description):
    outputs = {}
    for name in description["outputs"]:
        dtype = description["program"][name]["data_type"].type
        outputs[name] = helper.aligned(np.zeros(description["dimensions"], dtype=dtype), 64)
    return outputs


def _call_arguments(inputs, outputs):
    """Keyword arguments of the compiled program: arrays under `<name>_host`,
    0-D inputs under their bare name (reference run_program.py:164-169)."""
    kwargs = {}
    for group in (inputs, outputs):
        for name, value in group.items():
            is_array = getattr(value, "ndim", 0) > 0
            kwargs[name + "_host" if is_array else name] = value
    return kwargs


def _without_halo(arrays, halo):
    inner = slice(halo, -halo)
    return {name: a[(inner, ) * a.ndim] for name, a in arrays.items()}


def _dump(arrays, title, enabled):
    if enabled:
        for name, a in arrays.items():
            print(name + ":", a)


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
    log = _Log(log_level)
    description = helper.parse_json(stencil_file)
    name = _program_name(stencil_file)

    log("Creating kernel graph...")
    chain = KernelChainGraph(path=stencil_file, plot_graph=plot, log_level=log_level)

    if run_simulation:
        raise NotImplementedError(
            "The cycle-level FPGA simulator is not part of the HIP backend")
    if mode not in _MODES:
        raise ValueError("Unrecognized execution mode: {}".format(mode))
    if compare_to_reference and _REFERENCE_BACKEND is None:
        raise RuntimeError(
            "compare_to_reference needs a CPU checker: register one with "
            "stencilflow_amd.run_program.set_reference_backend()")

    log("Generating and compiling HIP kernels...")
    program = compile_program(chain, device=device, options=options)
    try:
        log(program.plan.describe(), LogLevel.MODERATE)
        if skip_execution or repetitions == 0:
            log("Skipping execution and exiting.")
            return None

        log("Loading input arrays...")
        directory = input_directory if input_directory is not None else os.path.dirname(stencil_file)
        inputs = _materialise_inputs(chain, description, directory, generate_input)
        log("Initializing output arrays...")
        outputs = _zeroed_outputs(description)

        kwargs = _call_arguments(inputs, outputs)
        for rep in range(repetitions):
            log("Executing program on the GPU..." if repetitions == 1 else
                "Executing repetition {}/{}...".format(rep + 1, repetitions))
            program(**kwargs)
            log("Finished running program.")
    finally:
        program.close()
    _dump(outputs, "result", print_result)

    expected = None
    if compare_to_reference:
        log("Executing reference program...")
        expected = _REFERENCE_BACKEND(stencil_file, inputs)
        log("Finished running program.")
        _dump(expected, "reference", print_result)

    if halo > 0:  # prune the shrink halo before saving / comparing (reference :202-209)
        outputs = _without_halo(outputs, halo)
        if expected is not None:
            expected = _without_halo(expected, halo)

    folder = os.path.join("results", name)
    os.makedirs(folder, exist_ok=True)
    helper.save_output_arrays(outputs, folder)
    log("Results saved to " + folder)
    if expected is None:
        return None

    reference_folder = os.path.join(folder, "reference")
    os.makedirs(reference_folder, exist_ok=True)
    helper.save_output_arrays(expected, reference_folder)
    log("Reference results saved to " + reference_folder)

    log("Comparing to reference...")
    for out_name, got in outputs.items():
        want = expected[out_name]
        if not helper.arrays_match(np.ravel(want), np.ravel(got), tolerance):
            print("Expected: {}".format(want))
            print("Got:      {}".format(got))
            raise ValueError("Result mismatch.")
    log("Results verified.")
    return 0
