import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mjx():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def data_dir():
    return os.path.join(ROOT, "tests", "data")


@pytest.fixture(scope="session")
def gpu_ctx(mjx):
    ctx = mjx.Context(0)
    yield ctx
    ctx.close()
