"""GPU: hotspot chains (centre-only auxiliary field per stage) with and without early
auxiliary-row requests (k1.auxpre)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stencilflow_amd as sf
from stencilflow_amd.backend import Plan
from stencilflow_amd.lowering import lower
import os, tempfile
from stencilflow_amd import programs
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp()
cases = []
for name, pos in (("hot3", ("float32", 16, 0.0, 512, 512, 512, 1, 1, 1)), ("hot2", ("float32", 16, 0.0, 0, 4096, 4096, 0, 1, 1))):
    prog, _ = programs.synthesize(*pos, stencil_shape="hotspot")
    cases.append((programs.write_program(prog, os.path.join(tmp, name + ".json")), tuple(prog["dimensions"])))
for f, shape in cases:
    sfir = lower(sf.KernelChainGraph(f))
    base = None
    for o in ("k1.auxpre=0", "k1.auxpre=1", "k1.auxpre=2", "k1.auxpre=0", "k1.auxpre=1", "k1.auxpre=2"):
        plan = Plan(sfir, options=o)
        plan.set_scalars([0.1] * len(plan.scalar_names))
        ins = [rng.random(shape, dtype=np.float32) for _ in plan.input_names] if base is None else ins
        plan.upload(ins)
        for _ in range(2): plan.execute(1); plan.synchronize()
        ts = []
        for _ in range(3):
            plan.execute(8); plan.synchronize(); ts.append(plan.elapsed_ms()/8)
        out = np.zeros(shape, np.float32); plan.download([out])
        same = "base" if base is None else ("same" if np.array_equal(out, base) else "DIFF")
        if base is None: base = out
        print(f.split("/")[-1], o, round(float(np.median(ts)),3), "ms", round(np.prod(shape)*16/np.median(ts)/1e3), "Mcells/s", same, flush=True)
        plan.close()
