import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the `-m gpu` suite: the oracle / fixture parity tests collect first, then the kernel-level parity files, and the
# engine / diagnostics files (self-consistency, process mode) last -- a failure late in the run must not hide the parity
# results (round 3: `pytest -x` stopped at test 54 of 241 and 187 parity tests never ran).  Files not listed keep their
# alphabetical place between the two groups; the order inside a file is untouched (sort is stable).
_FIRST = ["test_wkv7_gpu", "test_ref_kernel_gpu", "test_model_gpu", "test_parity_r3_gpu", "test_parity_r4_gpu",
          "test_worker_gpu", "test_fullsize_gpu", "test_mm8_spmv_gpu", "test_skinny_gemm_gpu", "test_fused_gpu"]
_LAST = ["test_diagnostics_gpu", "test_engine_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if stem in _FIRST:
            return _FIRST.index(stem)
        if stem in _LAST:
            return 1000 + _LAST.index(stem)
        return 500

    items.sort(key=rank)


@pytest.fixture(scope="session")
def oracle():
    from oracle import native

    native.build()
    return native
