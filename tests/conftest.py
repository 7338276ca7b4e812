import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rtmi():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def rtcheck():
    import rtcheck as m
    m.oracle_lib()
    return m


@pytest.fixture(scope="session")
def scenes_dir():
    return SCENES


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
