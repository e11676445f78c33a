import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")   # glibc's abort messages to stderr, not to /dev/tty


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/*.cpp), built on demand.  Test-only."""
    from oracle import oracle as O
    O.build()
    return O
