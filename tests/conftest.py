import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run with -m gpu on the MI355X box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Collection order of the GPU suite (it is run with -x): kernel parity first, then the engine against the oracle
# (with the full-size checks), the drop-in goldens, the other models, several ranks - and the tests that start fresh
# processes last, so that a process-level fault cannot hide a parity test.
_ORDER = ["test_kernels_gpu", "test_engine_gpu", "test_dropin_gpu", "test_cwgan_gpu", "test_curriculum_gpu", "test_ddp_gpu"]
_LAST = ["test_zz_no_alloc", "test_zzz_process_level"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _ORDER:
            return _ORDER.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 100
    items.sort(key=rank)          # stable: the order inside a file is kept
