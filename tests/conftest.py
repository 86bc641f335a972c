import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # packed sketch images (poppunk_amd/h5bulk.py) go to the user's cache directory: the suite gets its own
    import atexit
    import shutil
    import tempfile
    cache = tempfile.mkdtemp(prefix="ppk_test_cache_")
    atexit.register(shutil.rmtree, cache, ignore_errors=True)
    os.environ["XDG_CACHE_HOME"] = cache
    os.environ.pop("PPK_SIDECAR_DIR", None)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)          # shared checkers (tests/refine_golden.py)


@pytest.fixture
def ppk_option():
    """Set libppk_hip.so run-time options (ppk_set_option) for one test; restored afterwards."""
    from poppunk_amd import _lib
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)

    yield set_
    for name, value in saved.items():
        _lib.set_option(name, value)
