import os
import sys

import pytest

# torch bundles its own HIP runtime (same soname as /opt/rocm's).  Whichever is loaded first
# serves the whole process, and torch only finds its GPUs when its own copy won -- so tests that
# use both import torch BEFORE the HIP library is dlopen()ed (bench.py does the same).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The checker: CPU restatement (and the compiled reference when present)."""
    from oracle import oracle as O
    if not os.path.exists(O.ORACLE_SO):
        O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    import json

    def load(name):
        with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
            return json.load(f)
    return load
