#!/usr/bin/env python3
"""GPU: in-kernel cycle shares of the star kernel (diagnostic build, option
stamp=1).  Read the SHARES, not the total: the stamps serialise the stream."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stencilflow_amd as sf  # noqa: E402
from stencilflow_amd import programs  # noqa: E402
from stencilflow_amd.backend import Plan  # noqa: E402
from stencilflow_amd.lowering import lower  # noqa: E402

opts = sys.argv[1:] or ["-"]
with tempfile.TemporaryDirectory() as tmp:
    path = programs.write_program(programs.jacobi3d((512, 512, 512), 8), os.path.join(tmp, "p.json"))
    sfir = lower(sf.KernelChainGraph(path))
x = np.random.default_rng(0).random((512, 512, 512), dtype=np.float32)
for o in opts:
    opt = "stamp=1" + ("" if o == "-" else ";" + o)
    plan = Plan(sfir, options=opt)
    plan.upload([x])
    plan.execute(1); plan.synchronize(); plan.debug_counters()
    plan.execute(1); plan.synchronize()
    c = plan.debug_counters()
    tot = sum(c[:4])
    names = ["publish+barrier", "stage1(+input wait)", "load issue", "later stages(+barrier if !db)"]
    print(o, "ms=%.3f" % plan.elapsed_ms(), "waves=%d" % c[4], "cycles/wave=%.0f" % (tot / max(c[4], 1)),
          " | ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, c[:4])))
    plan.close()
