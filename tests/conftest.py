"""pytest configuration: registers the `gpu` marker and builds the oracle once.

`-m "not gpu"` : oracle vs golden fixtures / reference object code, host logic,
                 C-ABI symbol checks (no compute on a GPU).
`-m gpu`       : parity of the HIP path (through the C-ABI) against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def mn_bytes():
    with open(os.path.join(GOLDEN, "r0c1de5e1t_3_5.mn"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def octx32(oracle):
    """createCompressionContext(32, 8, 3.5) of the oracle (about 1 s)."""
    return oracle.OracleContext(32, 8, 3.5)
