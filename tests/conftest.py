import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the checker is compiled here, before any test can have touched the GPU (oracle_lib.lib() itself never builds)
    import oracle_lib
    if not os.path.exists(os.path.join(oracle_lib.ORACLE_DIR, "liboracle.so")):
        oracle_lib.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()          # raises if liboracle.so is missing (built at configure time / by build())
    return oracle_lib


@pytest.fixture(scope="session")
def hiplib():
    """The product library; built in-tree by __graft_entry__.build()."""
    from bayeslogit_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu(hiplib):
    import torch
    if hiplib.bl_device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("gpu-marked test on a machine without a HIP device")
    return torch.device("cuda:0")
