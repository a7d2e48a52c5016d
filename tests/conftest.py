import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize: a gpu test at BASELINE.json's full sizes that also spends about a minute in the CPU oracle "
                                       "(part of `-m gpu`; deselect locally with -m 'gpu and not fullsize')")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (oracle/libbx_oracle.so), built on demand.  Test infrastructure only."""
    from oracle import oracle_lib

    oracle_lib.build()
    return oracle_lib.lib()
