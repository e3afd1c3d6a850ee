import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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
