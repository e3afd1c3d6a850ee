#!/usr/bin/env python3
"""Host-side scaling of the C/OpenMP oracle (which thread count is the honest
CPU baseline on this box?)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    from stencilflow_amd import programs
    shape = (512, 512, 512)
    ref = c_oracle.CompiledReference(programs.jacobi3d(shape, 8))
    x = np.random.default_rng(0).random(shape, dtype=np.float32)
    x = ref.run({"a": x})["b7"]
    t = time.perf_counter()
    n = 0
    while time.perf_counter() - t < 4.0:
        x = ref.run({"a": x})["b7"]
        n += 8
    dt = time.perf_counter() - t
    print("threads=%s  %.0f Mcells/s" % (os.environ.get("OMP_NUM_THREADS"), 512**3 * n / dt / 1e6))
else:
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError as e:
        print("cgroup cpu.max unavailable", e)
    for t in sys.argv[1:] or ["8", "16", "32", "64", "128", "256"]:
        env = dict(os.environ, OMP_NUM_THREADS=t, OMP_PROC_BIND="close", OMP_PLACES="cores")
        subprocess.run([sys.executable, __file__, "child"], env=env)
