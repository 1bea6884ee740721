"""The differential fuzzer's plan generator (tools/fuzz_plans.py) must keep producing plans the ORACLE can execute — it is the
checker of every fuzzing run on the GPU box, and a generator that raises (or an oracle that cannot run what it generates) would
turn those runs into no-ops. CPU only: a few dozen seeds through the oracle, every operator kind reached, and the round-4 string
expressions (CASE that yields Utf8, LIKE with a pattern column) among them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_plans_run_through_the_oracle(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import fuzz_plans as fz
    finally:
        sys.path.pop(0)
    kinds, string_projections, executed = set(), 0, 0
    for seed in range(1, 61):
        rng = np.random.default_rng(seed)
        plan, _unordered = fz.random_plan(rng)
        kinds.add(type(plan).__name__)
        if type(plan).__name__ == "Projection":
            string_projections += sum(1 for e in plan.exprs if "CaseExpr" in type(e).__name__ or "Like" in type(e).__name__)
        try:
            oracle.execute(plan)
            executed += 1
        except oracle.OracleError:
            pass                      # a data-dependent error the reference reports too (divide by zero, overflow): compared by message on the GPU box
    assert {"HashAggregate", "HashJoinExec", "Filter", "Projection", "Sort", "Limit"} <= kinds, kinds
    assert string_projections > 0 and executed >= 40, (string_projections, executed)
