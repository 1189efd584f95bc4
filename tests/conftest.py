import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """Build the HIP library and the oracle's C restatement exactly as __graft_entry__.build() does.
    hipcc cross-compiles without a GPU; if it is absent the tests that need the library fail loudly
    on their own."""
    # always: make's dependencies make it a no-op when the library is current, and an edited .hip file
    # can never be tested against a stale libgpx.so (the library is git-ignored)
    try:
        import __graft_entry__
        __graft_entry__.build()
    except Exception as exc:           # report loudly; the dependent tests then fail with the real reason
        sys.stderr.write("\n*** conftest: building libgpx.so / the oracle FAILED: %r ***\n" % (exc,))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden
