import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as ge
    return ge.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_pkg()


@pytest.fixture(scope="session")
def engine_factory(pkg):
    made = []

    def make(**options):
        e = pkg.Engine(0, **options)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()
