"""The C++ host mirror of the reference's operator interface (include/qhip_plan.hpp): it must build against the C ABI
without a GPU, and on an MI355X its transcription of the reference's own unit tests (tests/cpp/mirror_tests.cpp) must pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    subprocess.check_call(["make", "-s", "-C", CPP])
    assert os.path.exists(os.path.join(CPP, "mirror_tests"))


def test_cpp_mirror_builds_against_the_c_abi():
    _build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_host"])   # the metric's step through the same mirror
    assert os.path.exists(os.path.join(ROOT, "tools", "bench_host"))
    with open(os.path.join(ROOT, "include", "qhip_plan.hpp")) as f:
        text = f.read()
    for node in ("Scan", "Filter", "Projection", "HashAggregate", "NoGroupingAggregate", "HashJoinExec", "NestedLoopJoinExec", "CrossJoin", "Sort", "Limit"):
        assert f"struct {node} :" in text, node


@pytest.mark.gpu
def test_cpp_mirror_runs_the_reference_unit_tests():
    _build()
    out = subprocess.run([os.path.join(CPP, "mirror_tests")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 failures" in out.stdout and "FAIL" not in out.stderr
