"""pytest configuration: markers and import paths.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks, gloo tests.
`-m gpu` runs on an MI355X box: parity of the HIP path (through the C-ABI) against the oracle.
"""
from __future__ import annotations

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG_PARENT = ROOT / "vectordb-retrieval_amd"
for p in (str(ROOT), str(PKG_PARENT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (C restatement, built on demand with gcc)."""
    from oracle import c_oracle

    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def vdb():
    """The product package (GPU tests only: importing it does not touch the GPU, using it does)."""
    import vdbhip

    return vdbhip
