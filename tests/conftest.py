"""pytest configuration: `gpu` marker (tests that need a real MI355X) and shared helpers.

CPU run (driver, every round):   python -m pytest tests/ -x -q -m "not gpu"
GPU run (driver, round end):     python -m pytest tests/ -x -q -m gpu      (one process, through the C ABI)
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import qurious_amd  # noqa: E402  (loads libqhip.so before anything may import torch)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import qoracle
    qoracle.lib()
    return qoracle


@pytest.fixture(scope="session")
def ctx():
    """The process-wide HIP context; GPU tests fail loudly (no skip, no fallback) when it cannot be created."""
    return qurious_amd.get_context()
