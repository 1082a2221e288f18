import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The oracle's C restatement is OpenMP code.  A GPU box shows every core of its host (256) but
# grants one GPU's share of them (16): without a cap each oracle call starts 256 threads on 16
# cores and slows down by orders of magnitude.  Must be set before libgomp is first loaded.
try:
    _cores = len(os.sched_getaffinity(0))
except AttributeError:
    _cores = os.cpu_count() or 1
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, _cores))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zk():
    return importlib.import_module("zk-state-proofs_amd")


@pytest.fixture(scope="session")
def fx():
    return importlib.import_module("zk-state-proofs_amd.fixtures")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def built_lib():
    build = importlib.import_module("zk-state-proofs_amd.build")
    return build.build()


@pytest.fixture(scope="session")
def host_client(zk, built_lib):
    """Verifier/executor-only client: never touches a GPU."""
    return zk.ProverClient(device=-1)


@pytest.fixture(scope="session")
def gpu(zk, built_lib):
    from util import Gpu
    return Gpu(zk)
