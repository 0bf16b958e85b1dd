"""pytest configuration: markers and import paths.

`-m "not gpu"` covers the oracle against the golden fixtures, the host logic and
the C-ABI's symbol table; `-m gpu` are the parity tests proper (HIP path through
the C ABI versus the oracle).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fusion-sim_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session", autouse=True)
def built_artefacts():
    """The HIP library, the N-API addon and the oracle are built artefacts kept out of git.  They
    normally travel with the working tree; on a bare checkout build them once (hipcc cross-compiles
    gfx950 without a GPU)."""
    needed = [os.path.join(ROOT, "fusion-sim_amd", "lib", "libfusionpic.so"),
              os.path.join(ROOT, "oracle", "libpic_oracle.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
