#!/bin/bash
# CPU: compiles the kernels the GPU tests will ask for into the in-tree code cache (.sf_cache/, git-ignored, travels
# with the snapshot).  Plan creation needs no device: every `-m gpu` test is run here, creates its plans -- which
# compiles and caches their kernels -- and then FAILS at its first device call; the failures are the expected outcome
# and are not reported.  Tests that start other processes are left out (their workers would only time out).
# usage: bash tools/warm_test_cache.sh [workers, default 6]
cd "$(dirname "$0")/.."
export SF_HIP_CACHE_DIR=$PWD/.sf_cache
rm -rf .sf_cache  # (a fresh cache: what no test asks for any more is gone)
mkdir -p .sf_cache
n=${1:-6}
timeout 3000 python -m pytest tests -q -m gpu -n $n -p no:cacheprovider \
  --ignore=tests/test_distributed.py --ignore=tests/test_capi_slab.py --ignore=tests/test_bench_launcher.py \
  --deselect tests/test_gpu_parity.py::test_command_line_with_a_named_reference_checker > /tmp/warm_test_cache.log 2>&1
tail -1 /tmp/warm_test_cache.log
# the bench's single-GPU workloads and everything __graft_entry__.build() compiles
python - <<'PY'
import __graft_entry__
__graft_entry__.build()
import bench
from stencilflow_amd.backend import Plan
for name, stages in (("c3", 1000), ("c2", 1000), ("c5", 300), ("box", 16), ("wide", 16), ("cross3", 8), ("dense", 4), ("fork", 16)):
    wl = bench.make_workload(name, 0, stages)
    _, sfir = bench.lower_program(wl["prog"])
    Plan(sfir).close()
PY
# ... and what the CPU suite compiles (it passes here; 8 minutes from a cold cache, 2 from a warm one)
timeout 2400 python -m pytest tests -q -m "not gpu" -n $n -p no:cacheprovider > /tmp/warm_test_cache_cpu.log 2>&1
tail -1 /tmp/warm_test_cache_cpu.log
echo "$(ls .sf_cache | wc -l) code objects, $(du -sh .sf_cache | cut -f1)"
