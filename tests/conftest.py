"""pytest configuration: registers the `gpu` marker and puts the package dirs on sys.path.

`-m "not gpu"`  : oracle vs golden vectors, host plumbing, C-ABI symbol check (no GPU needed)
`-m gpu`        : parity tests proper -- HIP path (through the C-ABI) vs the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
