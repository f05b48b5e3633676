import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle is test infrastructure: build it (and oracle/_ref when the reference is mounted) once."""
    from oracle import oracle as O

    O.build(ref=True)
    yield
