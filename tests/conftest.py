import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Factory of bcftools_amd.engine.Context objects; closes them at session end."""
    from bcftools_amd import engine
    made = []

    def make(cfg):
        c = engine.Context(cfg)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
