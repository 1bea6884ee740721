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


class _OracleEngine:
    """The CPU oracle as an engine: execute(plan) / evaluate(expr, batch)."""
    name = "oracle"

    def __init__(self, qoracle):
        self._o = qoracle

    def execute(self, plan):
        return self._o.execute(plan)

    def evaluate(self, expr, batch):
        return self._o.evaluate(expr, batch)


class _HipEngine:
    """The product path as an engine: plans execute through libqhip's C ABI; an expression is evaluated the way the
    reference's Projection does it (projection.rs:27-46), over a one-batch MemoryTable."""
    name = "hip"

    def execute(self, plan):
        return plan.execute()

    def evaluate(self, expr, batch):
        import pyarrow as pa
        scan = qurious_amd.Scan(batch.schema, qurious_amd.MemoryTable.try_new(batch.schema, [batch]), None, None)
        out = qurious_amd.Projection(None, scan, [expr]).execute()
        return pa.concat_arrays([b.column(0) for b in out]) if len(out) != 1 else out[0].column(0)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        from oracle import qoracle
        qoracle.lib()
        return _OracleEngine(qoracle)
    qurious_amd.get_context()   # fails loudly without a gfx950 device: no fallback
    return _HipEngine()


@pytest.fixture(params=["auto", "regions", "hashed", "dense", "dense_lds"])
def join_layout(request, monkeypatch):
    """Every join test runs under each table layout of csrc/join.cpp: the one libqhip picks by itself (`auto`: the dense
    direct-address layout for one integer key of a small value range, else one hashed table filled with atomics / the
    LDS-staged region build for big build sides), the region build forced on every join (QHIP_JOIN_REGION=2) and the hashed
    layouts only (QHIP_JOIN_DENSE=0), the dense layout wherever the key qualifies whatever the range-to-rows ratio
    (QHIP_JOIN_DENSE=2), and the dense layout with its bitmap staged in LDS whatever the pays-rule says (QHIP_JOIN_DENSE_LDS=2).
    (The byte-map form of the dense build, QHIP_JOIN_DENSE_BYTEMAP=1 — built in round 4, measured slower, off by default — is a
    layout of tests/test_gpu_dense_join.py.)"""
    if request.param == "regions":
        monkeypatch.setenv("QHIP_JOIN_REGION", "2")
        monkeypatch.setenv("QHIP_JOIN_DENSE", "0")
    elif request.param == "hashed":
        monkeypatch.setenv("QHIP_JOIN_DENSE", "0")
    elif request.param in ("dense", "dense_lds"):
        monkeypatch.setenv("QHIP_JOIN_DENSE", "2")
        if request.param == "dense_lds":
            monkeypatch.setenv("QHIP_JOIN_DENSE_LDS", "2")
    return request.param
