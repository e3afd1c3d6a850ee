"""ctypes binding of ``libsf_hip.so`` (C ABI: include/sf_hip.h).

``compile_program(chain)`` plays the role of ``sdfg.compile()`` in the
reference driver (stencilflow/run_program.py:118-128): it returns a callable
``program(**kwargs)`` taking arrays keyed ``<name>_host`` and 0-D inputs keyed
``<name>`` (run_program.py:164-169), operating in place on caller-owned NumPy
buffers, synchronously.

There is no CPU fallback: if the library cannot be loaded, or no GPU is
present when a program is run, a ``RuntimeError`` is raised.
"""

import ctypes
import os

import numpy as np

from .lowering import lower

_LIB = None
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                         os.environ.get("SF_HIP_LIBNAME", "libsf_hip.so"))

SF_OK = 0
_STATUS_EXC = {
    -1: ValueError,  # SF_ERR_INVALID
    -2: ValueError,  # SF_ERR_UNSUPPORTED
    -3: RuntimeError,  # SF_ERR_COMPILE
    -4: RuntimeError,  # SF_ERR_DEVICE
    -5: RuntimeError,  # SF_ERR_STATE
}

# every symbol include/sf_hip.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
_PP = ctypes.POINTER(ctypes.c_void_p)
_I = ctypes.c_int
_IP = ctypes.POINTER(ctypes.c_int)
_DP = ctypes.POINTER(ctypes.c_double)
_S = ctypes.c_char_p
_Z = ctypes.c_size_t
API = [
    ("sf_version", _I, []),
    ("sf_last_error", _S, []),
    ("sf_device_count", _I, []),
    ("sf_plan_create", _I, [_S, _I, _S, _PP]),
    ("sf_plan_destroy", _I, [_P]),
    ("sf_code_cache_stats", _I, [ctypes.POINTER(ctypes.c_long)] * 3 + [_I]),
    ("sf_self_checks_run", ctypes.c_long, []),
    ("sf_plan_kernel_verdict", _I, [_P, _I]),
    ("sf_plan_num_inputs", _I, [_P]),
    ("sf_plan_num_scalars", _I, [_P]),
    ("sf_plan_num_outputs", _I, [_P]),
    ("sf_plan_input_name", _S, [_P, _I]),
    ("sf_plan_scalar_name", _S, [_P, _I]),
    ("sf_plan_output_name", _S, [_P, _I]),
    ("sf_plan_input_bytes", _Z, [_P, _I]),
    ("sf_plan_output_bytes", _Z, [_P, _I]),
    ("sf_plan_set_scalars", _I, [_P, _DP, _I]),
    ("sf_plan_run", _I, [_P, _PP, _PP, _I]),
    ("sf_plan_upload", _I, [_P, _PP]),
    ("sf_plan_execute", _I, [_P, _I]),
    ("sf_plan_synchronize", _I, [_P]),
    ("sf_plan_download", _I, [_P, _PP]),
    ("sf_plan_elapsed_ms", _I, [_P, _DP]),
    ("sf_plan_num_launches", _I, [_P]),
    ("sf_plan_num_kernels", _I, [_P]),
    ("sf_plan_kernel_name", _S, [_P, _I]),
    ("sf_plan_kernel_source", _S, [_P, _I]),
    ("sf_plan_kernel_stats", _I, [_P, _I, _IP, _DP, _DP, _DP]),
    ("sf_plan_set_profile", _I, [_P, _I]),
    ("sf_plan_kernel_launch_times", _I, [_P, _I, _DP, _DP, _DP]),
    ("sf_plan_kernel_object", _I, [_P, _I, _PP, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_char_p)]),
    ("sf_plan_kernel_planes", _I, [_P, _I, _DP]),
    ("sf_plan_kernel_resources", _I, [_P, _I, _IP, _IP, _IP, _IP, _IP]),
    ("sf_describe_options", _S, []),
    ("sf_plan_describe", _S, [_P]),
    ("sf_plan_num_steps", _I, [_P]),
    ("sf_plan_step_halo", _I, [_P, _I, _IP, _IP]),
    ("sf_plan_step_inputs", _I, [_P, _I, _IP, _I]),
    ("sf_plan_step_output", _I, [_P, _I]),
    ("sf_plan_step_outputs", _I, [_P, _I, _IP, _I]),
    ("sf_plan_step_kernel", _I, [_P, _I]),
    ("sf_compiler_id", _S, []),
    ("sf_plan_execute_step", _I, [_P, _I, _I, _P]),
    ("sf_plan_execute_step_ranges", _I, [_P, _I, _I, _I, _I, _I, _P]),
    ("sf_plan_set_reserved_cus", _I, [_P, _I]),
    ("sf_plan_stream", _I, [_P, _PP]),
    ("sf_plan_num_buffers", _I, [_P]),
    ("sf_plan_buffer_info", _I,
     [_P, _I, _PP, ctypes.POINTER(ctypes.c_size_t), _IP]),
    ("sf_plan_input_buffer", _I, [_P, _I]),
    ("sf_plan_output_buffer", _I, [_P, _I]),
    ("sf_host_register", _I, [_P, _Z, _PP]),
    ("sf_host_unregister", _I, [_P]),
    ("sf_copy_async", _I, [_P, _P, _Z, _P]),
    ("sf_flag_set", _I, [_P, _P, ctypes.c_uint]),
    ("sf_flag_wait", _I, [_P, _P, ctypes.c_uint, ctypes.c_uint, _P]),
    ("sf_halo_create", _I, [_I, _I, _S, _I, ctypes.c_uint, _PP]),
    ("sf_halo_destroy", _I, [_P]),
    ("sf_halo_export", _I, [_P, _I, _P, _Z, _I, _I, _P]),
    ("sf_halo_connect", _I, [_P, _I, _P, _P]),
    ("sf_halo_start", _I, [_P, _I, _I, _P]),
    ("sf_halo_finish", _I, [_P, _I, _P]),
    ("sf_halo_check", _I, [_P]),
    ("sf_halo_fail", _I, [_P]),
    ("sf_halo_abandon", _I, [_P]),
    ("sf_halo_rccl_id", _I, [_P]),
    ("sf_halo_use_rccl", _I, [_P, _P, _I, _I]),
    ("sf_halo_transport", _S, [_P]),
    ("sf_halo_configure", _I, [_P, _I, _I]),
    ("sf_halo_set_profile", _I, [_P, _I]),
    ("sf_halo_exchange_times", _I, [_P, _IP, _DP, _DP]),
    ("sf_plan_execute_decomposed", _I, [_P, _P, _I]),
]
HALO_BLOB_BYTES = 256
HALO_RCCL_ID_BYTES = 128


def library_path():
    return _LIB_PATH


def describe_options():
    """The plan options the library accepts (``key=<value>  meaning`` per line); any other key is refused."""
    return load_library().sf_describe_options().decode()


def load_library():
    """Load ``libsf_hip.so``; fail loudly if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 /
    # libhiprtc (same sonames as /opt/rocm's).  If torch is going to be used in
    # this process (device streams, torch.distributed), its copy must be the one
    # that is loaded first, otherwise two runtimes coexist and the second sees
    # no GPU.  libsf_hip.so binds by soname and so shares whichever is resident.
    # $SF_HIP_COMGR pins the device compiler: hipRTC binds libamd_comgr by soname, so the copy that is
    # loaded FIRST compiles every kernel of the process (PyTorch brings its own).  Loading the named file
    # here, globally and before torch, makes it that copy; the library refuses to plan if it is not.
    pinned = os.environ.get("SF_HIP_COMGR")
    if pinned:
        try:
            ctypes.CDLL(pinned, mode=ctypes.RTLD_GLOBAL)
        except OSError as exc:
            raise RuntimeError("SF_HIP_COMGR={}: {}".format(pinned, exc))
    if os.environ.get("SF_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.isfile(_LIB_PATH):
        raise RuntimeError(
            "{} is missing: build it with `python -m stencilflow_amd.csrc.build`"
            " (there is no CPU fallback)".format(_LIB_PATH))
    try:
        lib = ctypes.CDLL(_LIB_PATH)
    except OSError as exc:
        raise RuntimeError("cannot load {}: {}".format(_LIB_PATH, exc))
    for name, restype, argtypes in API:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def _check(status):
    if status >= 0:
        return status
    msg = load_library().sf_last_error().decode()
    raise _STATUS_EXC.get(status, RuntimeError)(msg)


def code_cache_stats(drop_process_level=False):
    """(objects taken from the on-disk cache, objects compiled, cached objects the
    loader rejected and that were rebuilt) in this process so far."""
    vals = [ctypes.c_long(0) for _ in range(3)]
    _check(load_library().sf_code_cache_stats(*[ctypes.byref(v) for v in vals], 1 if drop_process_level else 0))
    return tuple(v.value for v in vals)


def _options_text(options):
    if not options:
        return None
    if isinstance(options, str):
        return options.encode()
    return ";".join("{}={}".format(k, v) for k, v in options.items()).encode()


class Plan:
    """Owner of one ``sf_plan`` handle."""

    def __init__(self, sfir_text, device=0, options=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        _check(
            self._lib.sf_plan_create(sfir_text.encode(), int(device),
                                     _options_text(options),
                                     ctypes.byref(self._h)))
        lib, h = self._lib, self._h
        self.input_names = [
            lib.sf_plan_input_name(h, i).decode()
            for i in range(lib.sf_plan_num_inputs(h))
        ]
        self.scalar_names = [
            lib.sf_plan_scalar_name(h, i).decode()
            for i in range(lib.sf_plan_num_scalars(h))
        ]
        self.output_names = [
            lib.sf_plan_output_name(h, i).decode()
            for i in range(lib.sf_plan_num_outputs(h))
        ]

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sf_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- introspection ----------------------------------------------------
    def describe(self):
        return self._lib.sf_plan_describe(self._h).decode()

    @property
    def num_launches(self):
        return self._lib.sf_plan_num_launches(self._h)

    def kernel_names(self):
        n = self._lib.sf_plan_num_kernels(self._h)
        return [self._lib.sf_plan_kernel_name(self._h, i).decode()
                for i in range(n)]

    def kernel_source(self, index):
        return self._lib.sf_plan_kernel_source(self._h, index).decode()

    def kernel_stats(self):
        """name -> dict(launches, total_ms, updates_per_launch,
        algorithmic_bytes_per_launch); timings need option ``profile=1``."""
        out = {}
        for i, name in enumerate(self.kernel_names()):
            n = ctypes.c_int()
            ms, upd, byt = (ctypes.c_double(), ctypes.c_double(),
                            ctypes.c_double())
            _check(
                self._lib.sf_plan_kernel_stats(self._h, i, ctypes.byref(n),
                                               ctypes.byref(ms),
                                               ctypes.byref(upd),
                                               ctypes.byref(byt)))
            out[name] = dict(launches=n.value,
                             total_ms=ms.value,
                             updates_per_launch=upd.value,
                             algorithmic_bytes_per_launch=byt.value)
        return out

    def kernel_verdicts(self):
        """name -> verdict of the plan-time self-check (0 not checked, 1 passed, 2 failed)."""
        return {name: self._lib.sf_plan_kernel_verdict(self._h, i) for i, name in enumerate(self.kernel_names())}

    def set_profile(self, on=True):
        """Per-launch HIP events on / off (resets the per-kernel counters)."""
        _check(self._lib.sf_plan_set_profile(self._h, 1 if on else 0))

    def kernel_planes(self):
        """name -> planes written by the profiled launches of that kernel."""
        out = {}
        for i, name in enumerate(self.kernel_names()):
            v = ctypes.c_double()
            _check(self._lib.sf_plan_kernel_planes(self._h, i, ctypes.byref(v)))
            out[name] = v.value
        return out

    def kernel_object(self, index):
        """(code object bytes, extra compiler flags) of kernel ``index``, as the plan loads it."""
        data, size, flags = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_char_p()
        _check(self._lib.sf_plan_kernel_object(self._h, index, ctypes.byref(data), ctypes.byref(size), ctypes.byref(flags)))
        return ctypes.string_at(data.value, size.value), (flags.value or b"").decode()

    def kernel_launch_times(self):
        """name -> (min, median, max) HIP-event duration in ms of the profiled launches of that
        kernel (kernels without a profiled launch are left out)."""
        out = {}
        for i, name in enumerate(self.kernel_names()):
            v = [ctypes.c_double() for _ in range(3)]
            if self._lib.sf_plan_kernel_launch_times(self._h, i, *[ctypes.byref(x) for x in v]) == 0:
                out[name] = tuple(x.value for x in v)
        return out

    def kernel_resources(self):
        """name -> dict(vgprs, agprs, spills, scratch, lds) from the code object."""
        out = {}
        for i, name in enumerate(self.kernel_names()):
            v = [ctypes.c_int() for _ in range(5)]
            _check(self._lib.sf_plan_kernel_resources(
                self._h, i, *[ctypes.byref(x) for x in v]))
            out[name] = dict(zip(("vgprs", "agprs", "spills", "scratch", "lds"),
                                 [x.value for x in v]))
        return out

    def input_bytes(self, i):
        return self._lib.sf_plan_input_bytes(self._h, i)

    def output_bytes(self, i):
        return self._lib.sf_plan_output_bytes(self._h, i)

    # -- execution --------------------------------------------------------
    def _ptr_array(self, arrays, sizes, what):
        n = len(sizes)
        if len(arrays) != n:
            raise ValueError("expected {} {} arrays, got {}".format(
                n, what, len(arrays)))
        ptrs = (ctypes.c_void_p * max(n, 1))()
        for i, a in enumerate(arrays):
            if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
                raise ValueError(
                    "{} array {} must be a C-contiguous ndarray".format(
                        what, i))
            if a.nbytes != sizes[i]:
                raise ValueError(
                    "{} array {} has {} bytes, the program expects {}".format(
                        what, i, a.nbytes, sizes[i]))
            ptrs[i] = a.ctypes.data
        return ptrs

    def set_scalars(self, values):
        arr = (ctypes.c_double * max(len(values), 1))(*[float(v)
                                                        for v in values])
        _check(self._lib.sf_plan_set_scalars(self._h, arr, len(values)))

    def run(self, inputs, outputs, repetitions=1):
        ins = self._ptr_array(
            inputs, [self.input_bytes(i) for i in range(len(self.input_names))],
            "input")
        outs = self._ptr_array(
            outputs,
            [self.output_bytes(i) for i in range(len(self.output_names))],
            "output")
        _check(self._lib.sf_plan_run(self._h, ins, outs, int(repetitions)))

    def upload(self, inputs):
        ins = self._ptr_array(
            inputs, [self.input_bytes(i) for i in range(len(self.input_names))],
            "input")
        _check(self._lib.sf_plan_upload(self._h, ins))

    def execute(self, repetitions=1):
        _check(self._lib.sf_plan_execute(self._h, int(repetitions)))

    def synchronize(self):
        _check(self._lib.sf_plan_synchronize(self._h))

    def download(self, outputs):
        outs = self._ptr_array(
            outputs,
            [self.output_bytes(i) for i in range(len(self.output_names))],
            "output")
        _check(self._lib.sf_plan_download(self._h, outs))

    def elapsed_ms(self):
        ms = ctypes.c_double()
        _check(self._lib.sf_plan_elapsed_ms(self._h, ctypes.byref(ms)))
        return ms.value

    # -- slab stepping (multi-GPU driver) -----------------------------------
    @property
    def num_steps(self):
        return self._lib.sf_plan_num_steps(self._h)

    def step_halo(self, step):
        buf, depth = ctypes.c_int(), ctypes.c_int()
        _check(
            self._lib.sf_plan_step_halo(self._h, step, ctypes.byref(buf),
                                        ctypes.byref(depth)))
        return buf.value, depth.value

    def step_inputs(self, step):
        # the call returns how many buffers the step reads: size the array from that
        n = _check(self._lib.sf_plan_step_inputs(self._h, step, None, 0))
        ids = (ctypes.c_int * max(1, n))()
        n = _check(self._lib.sf_plan_step_inputs(self._h, step, ids, n))
        return [ids[i] for i in range(n)]

    def step_output(self, step):
        return _check(self._lib.sf_plan_step_output(self._h, step))

    def step_outputs(self, step):
        """Buffers launch ``step`` writes (several for a DAG group)."""
        ids = (ctypes.c_int * 8)()
        n = _check(self._lib.sf_plan_step_outputs(self._h, step, ids, 8))
        return [ids[i] for i in range(n)]

    def step_kernel(self, step):
        """Index (into ``kernel_names()``) of the compiled kernel launch ``step`` runs."""
        return _check(self._lib.sf_plan_step_kernel(self._h, step))

    def compiler(self):
        """The device compiler of this process (hipRTC + the libamd_comgr it binds, with versions)."""
        return (self._lib.sf_compiler_id() or b"?").decode()

    def execute_step(self, step, part=0, stream=None):
        _check(
            self._lib.sf_plan_execute_step(self._h, step, part,
                                           ctypes.c_void_p(stream or 0)))

    def execute_step_ranges(self, step, i_begin, i_end, i_begin2=0, i_end2=0,
                            stream=None):
        _check(
            self._lib.sf_plan_execute_step_ranges(self._h, step, int(i_begin),
                                                  int(i_end), int(i_begin2),
                                                  int(i_end2),
                                                  ctypes.c_void_p(stream or 0)))

    def set_reserved_cus(self, cus):
        """Leave `cus` compute units free in the launches that follow (0 = none)."""
        _check(self._lib.sf_plan_set_reserved_cus(self._h, int(cus)))

    @property
    def num_buffers(self):
        return _check(self._lib.sf_plan_num_buffers(self._h))

    def buffer_info(self, buffer_id):
        ptr = ctypes.c_void_p()
        plane = ctypes.c_size_t()
        planes = ctypes.c_int()
        _check(
            self._lib.sf_plan_buffer_info(self._h, buffer_id,
                                          ctypes.byref(ptr),
                                          ctypes.byref(plane),
                                          ctypes.byref(planes)))
        return ptr.value, plane.value, planes.value

    def input_buffer(self, i):
        return _check(self._lib.sf_plan_input_buffer(self._h, i))

    def output_buffer(self, i):
        return _check(self._lib.sf_plan_output_buffer(self._h, i))


class CompiledProgram:
    """``program(**kwargs)``: the compiled-program call of the reference driver
    (stencilflow/run_program.py:164-178)."""

    def __init__(self, chain, device=0, options=None):
        self.chain = chain
        self.sfir = lower(chain)
        self.plan = Plan(self.sfir, device=device, options=options)

    def __call__(self, **kwargs):
        plan = self.plan
        ins, outs, scalars = [], [], []
        for name in plan.input_names:
            key = name + "_host"
            if key not in kwargs:
                raise TypeError("missing argument '{}'".format(key))
            arr = kwargs[key]
            want = self.chain.inputs[name]["data_type"].type
            if not isinstance(arr, np.ndarray) or arr.dtype != want:
                raise TypeError("argument '{}' must be an ndarray of {}".format(
                    key, np.dtype(want).name))
            ins.append(np.ascontiguousarray(arr))
        for name in plan.scalar_names:
            if name not in kwargs:
                raise TypeError("missing scalar argument '{}'".format(name))
            scalars.append(float(kwargs[name]))
        for name in plan.output_names:
            key = name + "_host"
            if key not in kwargs:
                raise TypeError("missing argument '{}'".format(key))
            arr = kwargs[key]
            want = self.chain.program[name]["data_type"].type
            if not isinstance(arr, np.ndarray) or arr.dtype != want \
                    or not arr.flags["C_CONTIGUOUS"]:
                raise TypeError(
                    "argument '{}' must be a C-contiguous ndarray of {}".format(
                        key, np.dtype(want).name))
            outs.append(arr)
        if scalars:
            plan.set_scalars(scalars)
        plan.run(ins, outs, 1)

    def close(self):
        self.plan.close()


def compile_program(chain, device=0, options=None):
    return CompiledProgram(chain, device=device, options=options)
