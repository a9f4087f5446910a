import os
import sys

import pytest

try:  # load order: torch bundles its own HIP runtime; it must be in the process before libo3ds_backend.so pulls in /opt/rocm's,
    import torch  # noqa: F401  # otherwise a later torch.cuda init reports "No HIP GPUs are available" (seen with pytest -k on one test)
except Exception:  # torch is plumbing for the sharded path only; the backend itself does not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def small_c2():
    """Reduced config-2 inputs (8192-pt scan vs 100k-pt map), seeded."""
    from open3d_slam_amd import synthetic as syn

    return syn.config2_inputs(n_map=100_000, n_az=512)


@pytest.fixture(scope="session")
def backend_f32():
    from open3d_slam_amd import backend

    be = backend.Backend(0, backend.PRECISION_F32)
    yield be
    be.close()


@pytest.fixture(scope="session")
def backend_f64():
    from open3d_slam_amd import backend

    be = backend.Backend(0, backend.PRECISION_F64)
    yield be
    be.close()
