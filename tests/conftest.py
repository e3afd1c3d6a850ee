import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The tests keep their compiled kernels in the tree (git-ignored): hipRTC compiles without a GPU, so the code objects the
# GPU tests need can be made where there is none (tools/warm_test_cache.sh) and travel to the GPU box with the snapshot,
# like the built libsf_hip.so -- the suite then spends its minutes on running kernels, not on compiling them.  The cache
# is keyed by source, flags and the compiler in use (csrc/codecache.cpp), so a stale entry is a miss, never a wrong
# kernel; plan-time self-check verdicts are recorded on the GPU only.  Tests of the cache itself set their own directory.
os.environ.setdefault("SF_HIP_CACHE_DIR", os.path.join(ROOT, ".sf_cache"))

PROGRAMS = os.path.join(ROOT, "tests", "golden", "programs")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers",
                            "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def programs_dir():
    return PROGRAMS


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _register_reference_checker():
    """The CPU checker behind run_program(compare_to_reference=True) is the
    oracle; only the test-suite wires it in."""
    from tests.reference_provider import register
    register()
    yield
