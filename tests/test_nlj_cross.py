"""NestedLoopJoinExec / CrossJoin (physical/plan/join/{nest_loop_join,cross_join}.rs): oracle pinned
by the reference's unit-test vectors (CPU), HIP path vs those vectors and vs the oracle on random data (gpu)."""
from __future__ import annotations

import json
import os

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from oracle import qoracle
from qurious_amd import JoinSide, JoinType, Operator

from .helpers import build_table_scan_i32, col, rows_of, table_scan

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")
I32 = pa.int32()


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _nlj_case(case):
    left, right = build_table_scan_i32(case["left"]), build_table_scan_i32(case["right"])
    jf = None
    if case["filter"]:
        f = case["filter"]
        ln, rn = list(case["left"])[f["left_col"]], list(case["right"])[f["right_col"]]
        schema = pa.schema([pa.field(ln, I32, False), pa.field(rn, I32, False)])
        jf = q.JoinFilter(q.BinaryExpr(col(ln, 0), Operator[f["op"]], col(rn, 1)), [(f["left_col"], JoinSide.Left), (f["right_col"], JoinSide.Right)], schema)
    return q.NestedLoopJoinExec.try_new(left, right, JoinType[case["join_type"]], jf)


def test_oracle_nlj_and_cross_join_reference_vectors(golden):
    for case in golden["nested_loop_join_exec"]["cases"]:
        assert rows_of(qoracle.execute(_nlj_case(case))) == [tuple(r) for r in case["expected"]], case["name"]
    cj = golden["cross_join_exec"]
    out = qoracle.execute(q.CrossJoin(build_table_scan_i32(cj["left"]), build_table_scan_i32(cj["right"])))
    assert len(out) == cj["n_batches"] and rows_of(out) == [tuple(r) for r in cj["expected"]]


def test_planner_picks_nested_loop_join_without_keys():
    a, b = build_table_scan_i32({"x": [1]}), build_table_scan_i32({"y": [2]})
    p = q.DefaultQueryPlanner()
    assert isinstance(p.physical_plan_join(a, b, JoinType.Inner, [], None), q.NestedLoopJoinExec)
    assert isinstance(p.physical_plan_join(a, b, JoinType.Inner, [(col("x", 0), col("y", 0))], None), q.HashJoinExec)
    assert isinstance(p.physical_plan_cross_join(a, b), q.CrossJoin)


def _same(got, want):
    assert [b.num_rows for b in got] == [b.num_rows for b in want]
    assert rows_of(got) == rows_of(want)


def _sides(rng, nl, nr):
    def side(n, p):
        a = pa.array(rng.integers(0, 12, n), type=pa.int64(), mask=rng.random(n) < 0.1)
        s = pa.array([("v%d" % v) * (1 + v % 3) for v in rng.integers(0, 9, n)], type=pa.string(), mask=rng.random(n) < 0.1)
        schema = pa.schema([pa.field(p + "k", pa.int64(), True), pa.field(p + "s", pa.string(), True)])
        return schema, pa.RecordBatch.from_arrays([a, s], schema=schema)
    return side(nl, "l_"), side(nr, "r_")


@pytest.mark.gpu
def test_gpu_nlj_and_cross_join_reference_vectors(golden):
    q.get_context()
    for case in golden["nested_loop_join_exec"]["cases"]:
        plan = _nlj_case(case)
        got = plan.execute()
        assert rows_of(got) == [tuple(r) for r in case["expected"]], case["name"]
        _same(got, qoracle.execute(plan))
    cj = golden["cross_join_exec"]
    plan = q.CrossJoin(build_table_scan_i32(cj["left"]), build_table_scan_i32(cj["right"]))
    got = plan.execute()
    assert len(got) == cj["n_batches"] and rows_of(got) == [tuple(r) for r in cj["expected"]]


@pytest.mark.gpu
def test_gpu_nlj_all_types_vs_oracle():
    q.get_context()
    rng = np.random.default_rng(12)
    (ls, lb), (rs, rb) = _sides(rng, 90, 70)
    left = table_scan(ls, [lb.slice(0, 40), lb.slice(40, 50)])
    right = table_scan(rs, [rb.slice(0, 1), rb.slice(1, 69)])
    fschema = pa.schema([pa.field("l_k", pa.int64()), pa.field("r_k", pa.int64())])
    lt = q.JoinFilter(q.BinaryExpr(col("l_k", 0), Operator.Lt, col("r_k", 1)), [(0, JoinSide.Left), (0, JoinSide.Right)], fschema)
    never = q.JoinFilter(q.BinaryExpr(col("l_k", 0), Operator.Gt, q.BinaryExpr(col("r_k", 1), Operator.Add, q.Literal(q.ScalarValue.Int64(100)))),
                         [(0, JoinSide.Left), (0, JoinSide.Right)], fschema)
    empty_r = table_scan(rs, [rb.slice(0, 0)])
    empty_l = table_scan(ls, [lb.slice(0, 0)])
    for jt in JoinType:
        for l, r, f in ((left, right, lt), (left, right, None), (left, right, never), (left, empty_r, lt), (empty_l, right, lt), (empty_l, empty_r, None)):
            plan = q.NestedLoopJoinExec.try_new(l, r, jt, f)
            _same(plan.execute(), qoracle.execute(plan))


@pytest.mark.gpu
def test_gpu_cross_join_batches_vs_oracle():
    q.get_context()
    rng = np.random.default_rng(13)
    (ls, lb), (rs, rb) = _sides(rng, 7, 9)
    left = table_scan(ls, [lb.slice(0, 3), lb.slice(3, 0), lb.slice(3, 4)])
    right = table_scan(rs, [rb.slice(0, 4), rb.slice(4, 5)])
    plan = q.CrossJoin(left, right)
    got, want = plan.execute(), qoracle.execute(plan)
    assert len(got) == len(want) == 7 * 2
    _same(got, want)
    for l, r in ((left, table_scan(rs, [rb.slice(0, 0)])), (table_scan(ls, [lb.slice(0, 0)]), right)):
        plan = q.CrossJoin(l, r)
        _same(plan.execute(), qoracle.execute(plan))
    # a cross join feeding an aggregate (the shape scalar subqueries decorrelate into)
    agg = q.HashAggregate(None, plan if False else q.CrossJoin(left, right), [col("l_k", 0)], [q.CountAggregateExpr(col("r_k", 2))])
    assert sorted(rows_of(agg.execute()), key=repr) == sorted(rows_of(qoracle.execute(agg)), key=repr)
