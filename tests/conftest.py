import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle works on small gather / scatter-heavy tensors: torch's intra-op threading over every core of a
    256-core GPU host is many times SLOWER there than a handful of threads (bench.py: cpu_baseline measures it).  Cap it;
    TT_ORACLE_THREADS overrides."""
    import torch
    n = int(os.environ.get("TT_ORACLE_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Explicit build step (NOT a fallback): the HIP library is compiled in-tree with hipcc before any test uses it;
    a no-op when the source hash embedded in triplaneturbo_amd/libtt_hip.so matches the tree."""
    from triplaneturbo_amd import _lib
    if os.environ.get("TT_LIB_VARIANT"):  # dev only: run the suite against an experiment build (tools/build_variants.py)
        _lib.use_variant(os.environ["TT_LIB_VARIANT"])
    else:
        _lib.build()
    yield
