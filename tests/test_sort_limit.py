"""Sort / Limit (physical/plan/sort.rs, limit.rs). The CPU tests pin the oracle's restatement of
arrow's lexsort against the reference's own expected outputs (tests/golden/reference_vectors.json: sort.rs / limit.rs unit
tests, order_by.slt, limit.slt); the gpu tests hold the HIP path to the same vectors and to the oracle on random data."""
from __future__ import annotations

import decimal
import json
import math
import os

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from oracle import qoracle
from qurious_amd import ScalarValue as S

from .helpers import col, rows_of, table_scan

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")
I32 = pa.int32()
TYPES = {"int32": pa.int32(), "float64": pa.float64(), "uint64": pa.uint64()}


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _scan_of(columns) -> q.Scan:
    names = list(columns)
    schema = pa.schema([pa.field(n, TYPES[columns[n][0]], False) for n in names])
    batch = pa.RecordBatch.from_arrays([pa.array(columns[n][1], type=TYPES[columns[n][0]]) for n in names], schema=schema)
    return table_scan(schema, [batch])


def _t(rows, names=("v1", "v2")) -> q.Scan:
    schema = pa.schema([pa.field(n, I32, True) for n in names])
    arrays = [pa.array([r[k] for r in rows], type=I32) for k in range(len(names))]
    return table_scan(schema, [pa.RecordBatch.from_arrays(arrays, schema=schema)])


def _planner_sort(scan, keys):
    """keys = [(column index, asc)] the way the reference's planner lowers ORDER BY (nulls_first = true)"""
    return q.DefaultQueryPlanner().physical_plan_sort(scan, [(col(scan.schema().field(i).name, i), asc) for i, asc in keys])


def _golden_plans(golden):
    """(name, plan, expected rows) for every reference vector of the sort / limit path"""
    out = []
    for case in golden["sort_exec"]["cases"]:
        scan = _scan_of(case["columns"])
        exprs = [q.PhysicalSortExpr(col(list(case["columns"])[k["column"]], k["column"]), q.SortOptions(k["descending"], k["nulls_first"]))
                 for k in case["keys"]]
        out.append((case["name"], q.Sort.new_with_limit(exprs, scan, case["limit"]), [tuple(r) for r in case["expected"]]))
    lim = golden["limit_exec"]
    out.append(("test_limit", q.Limit(_scan_of(lim["columns"]), lim["fetch"], lim["skip"]), [tuple(r) for r in lim["expected"]]))
    ob = golden["slt"]["order_by"]
    out.append(("order by v1 asc", _planner_sort(_t(ob["t1"]), [(0, True)]), [tuple(r) for r in sorted(ob["t1"])]))
    out.append(("order by v1 desc", _planner_sort(_t(ob["t1"]), [(0, False)]), [tuple(r) for r in sorted(ob["t1"], reverse=True)]))
    out.append(("order by v1 asc, v2 desc", _planner_sort(_t(ob["t2"]), [(0, True), (1, False)]), [tuple(r) for r in ob["v1_asc_v2_desc"]]))
    out.append(("order by with NULLs", _planner_sort(_t(ob["t3"]), [(0, True), (1, True)]), [tuple(r) for r in ob["v1_asc_v2_asc_with_nulls"]]))
    li = golden["slt"]["limit"]
    rows = {r[0]: tuple(r) for r in li["t"]}
    for name, fetch, skip in (("limit_3", 3, 0), ("offset_2", None, 2), ("limit_2_offset_2", 2, 2), ("limit_6", 6, 0), ("limit_0", 0, 0),
                              ("offset_5", None, 5)):
        out.append((name, q.Limit(_t(li["t"]), fetch, skip), [rows[v] for v in li[name]]))
    return out


def test_order_by_slt_first_columns(golden):
    """the .slt files print v1 only for the single-key queries: check those transcriptions against the sorted rows"""
    ob = golden["slt"]["order_by"]
    assert [r[0] for r in sorted(ob["t1"])] == ob["v1_asc"]
    assert [r[0] for r in sorted(ob["t1"], reverse=True)] == ob["v1_desc"]


def test_oracle_sort_limit_against_reference_vectors(golden):
    for name, plan, expected in _golden_plans(golden):
        assert rows_of(qoracle.execute(plan)) == expected, name


def test_oracle_limit_batch_structure():
    """limit.rs:36-55: whole batches are skipped, and a window that closes exactly on a batch boundary with more batches
    behind it yields one empty slice before the loop breaks"""
    schema = pa.schema([pa.field("v", I32, False)])
    b = lambda vals: pa.RecordBatch.from_arrays([pa.array(vals, type=I32)], schema=schema)   # noqa: E731
    scan = table_scan(schema, [b([1, 2]), b([3, 4, 5]), b([6])])
    sizes = lambda plan: [x.num_rows for x in qoracle.execute(plan)]                       # noqa: E731
    assert sizes(q.Limit(scan, None, 0)) == [2, 3, 1]
    assert sizes(q.Limit(scan, 5, 0)) == [2, 3, 0]
    assert sizes(q.Limit(scan, 3, 2)) == [3, 0]
    assert sizes(q.Limit(scan, 2, 3)) == [2]      # (the one-row third batch is dropped by the skip = 1 that was never cleared)
    assert sizes(q.Limit(scan, 0, 0)) == [0]
    assert sizes(q.Limit(scan, None, 6)) == []
    # limit.rs:39-44: `skip` is never cleared after the batch it was applied to — OFFSET 4 skips [1, 2], takes 3 and 4 off the
    # second batch (leaving 5) and then, still holding skip = 2, drops the one-row third batch whole. (SQL would return 5, 6.)
    assert sizes(q.Limit(scan, 100, 4)) == [1]
    assert rows_of(qoracle.execute(q.Limit(scan, 100, 4))) == [(5,)]
    wide = table_scan(schema, [b([1, 2, 3]), b([4, 5, 6, 7]), b([8, 9, 10])])
    assert rows_of(qoracle.execute(q.Limit(wide, None, 1))) == [(2,), (3,), (5,), (6,), (7,), (9,), (10,)]   # every batch loses its first row


def test_oracle_lexsort_null_and_float_order():
    """arrow's sort: NULL placement is independent of `descending`; floats in total order (-NaN < -inf < -0 < +0 < inf < NaN)"""
    a = pa.array([3, None, 1, 2, None], type=I32)
    assert list(qoracle.lexsort_to_indices([(a, False, True)])) == [1, 4, 2, 3, 0]
    assert list(qoracle.lexsort_to_indices([(a, True, True)])) == [1, 4, 0, 3, 2]
    assert list(qoracle.lexsort_to_indices([(a, True, False)])) == [0, 3, 2, 1, 4]
    f = pa.array([0.0, -0.0, float("nan"), float("-inf"), 1.5, float("inf")], type=pa.float64())
    assert list(qoracle.lexsort_to_indices([(f, False, True)])) == [3, 1, 0, 4, 5, 2]


# ------------------------------------------------------------------------------------------------------------ gpu
def _exact(rows):
    """rows with floats made comparable bit for bit (NaN == NaN, -0.0 != 0.0)"""
    def norm(v):
        if isinstance(v, float):
            return ("nan",) if v != v else (v, math.copysign(1.0, v))
        return v
    return [tuple(norm(v) for v in r) for r in rows]


def _batches_equal(got, want):
    assert [b.num_rows for b in got] == [b.num_rows for b in want]
    assert _exact(rows_of(got)) == _exact(rows_of(want))


@pytest.mark.gpu
def test_gpu_sort_limit_reference_vectors(golden):
    q.get_context()
    for name, plan, expected in _golden_plans(golden):
        got = plan.execute()
        assert rows_of(got) == expected, name
        _batches_equal(got, qoracle.execute(plan))


def _random_table(rng, n, null_p=0.1):
    D = decimal.Decimal
    dec = pa.decimal128(15, 2)
    m = lambda: rng.random(n) < null_p   # noqa: E731
    fl = rng.normal(size=n)
    fl[rng.integers(0, n, max(1, n // 50))] = np.nan
    fl[rng.integers(0, n, max(1, n // 50))] = -0.0
    fl[rng.integers(0, n, max(1, n // 50))] = 0.0
    fl[rng.integers(0, n, max(1, n // 100))] = np.inf
    cols = {
        "i32": pa.array(rng.integers(-50, 50, n), type=pa.int32(), mask=m()),
        "i64": pa.array(rng.integers(-2**62, 2**62, n), type=pa.int64(), mask=m()),
        "dec": pa.array([D(int(v)).scaleb(-2) for v in rng.integers(-10**12, 10**12, n)], type=dec, mask=m()),
        "f64": pa.array(fl, type=pa.float64(), mask=m()),
        "s": pa.array([("k%d" % v) * (1 + v % 4) + ("\x00" if v % 7 == 0 else "") for v in rng.integers(0, 40, n)], type=pa.string(), mask=m()),
        "d": pa.array(rng.integers(8000, 8100, n), type=pa.int32(), mask=m()).cast(pa.date32()),
        "b": pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=m()),
        "u8": pa.array(rng.integers(0, 4, n), type=pa.uint8()),
        "row": pa.array(np.arange(n), type=pa.int64()),
    }
    schema = pa.schema([pa.field(k, v.type, True) for k, v in cols.items()])
    return schema, pa.RecordBatch.from_arrays(list(cols.values()), schema=schema)


@pytest.mark.gpu
def test_gpu_sort_every_key_type_vs_oracle():
    q.get_context()
    rng = np.random.default_rng(4242)
    schema, batch = _random_table(rng, 5000)
    cuts = [0, 1000, 1000, 3333, 5000]
    scan = table_scan(schema, [batch.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    names = [f.name for f in schema]
    for k, name in enumerate(names[:-1]):
        for desc in (False, True):
            for nf in (False, True):
                plan = q.Sort([q.PhysicalSortExpr(col(name, k), q.SortOptions(desc, nf))], scan)
                _batches_equal(plan.execute(), qoracle.execute(plan))
    # several keys, low-cardinality ones first so that the later keys and the implicit row-number key matter
    plan = q.Sort([q.PhysicalSortExpr(col("u8", 7), q.SortOptions(True, True)), q.PhysicalSortExpr(col("b", 6), q.SortOptions(False, False)),
                   q.PhysicalSortExpr(col("s", 4), q.SortOptions(True, False)), q.PhysicalSortExpr(col("i32", 0), q.SortOptions(False, True))], scan)
    _batches_equal(plan.execute(), qoracle.execute(plan))
    # an expression key and top-N
    expr = q.BinaryExpr(col("i32", 0), q.Operator.Mul, q.Literal(S.Int32(-3)))
    for limit in (0, 1, 17, 5000, 10**6):
        plan = q.Sort.new_with_limit([q.PhysicalSortExpr(expr, q.SortOptions(False, True)), q.PhysicalSortExpr(col("dec", 2), q.SortOptions(True, True))],
                                     scan, limit)
        _batches_equal(plan.execute(), qoracle.execute(plan))


@pytest.mark.gpu
def test_gpu_sort_and_limit_edge_cases():
    q.get_context()
    rng = np.random.default_rng(7)
    schema, batch = _random_table(rng, 300)
    empty = table_scan(schema, [batch.slice(0, 0)])
    none = q.Scan(schema, q.MemoryTable.try_new(schema, []))
    one = table_scan(schema, [batch.slice(5, 1)])
    key = [q.PhysicalSortExpr(col("i64", 1), q.SortOptions(False, True))]
    for src in (empty, none, one):
        for plan in (q.Sort(key, src), q.Sort([], src), q.Limit(src, 3, 0), q.Limit(src, None, 1)):
            _batches_equal(plan.execute(), qoracle.execute(plan))
    scan = table_scan(schema, [batch.slice(0, 100), batch.slice(100, 150), batch.slice(250, 50)])
    for fetch, skip in ((None, 0), (250, 0), (150, 100), (10, 95), (0, 0), (None, 300), (1000, 120), (None, 299), (None, 30), (140, 30), (None, 60)):
        plan = q.Limit(scan, fetch, skip)
        _batches_equal(plan.execute(), qoracle.execute(plan))
    # ORDER BY ... LIMIT n OFFSET m the way the planner lowers it (top-N sort of skip + fetch rows, then the window)
    plan = q.DefaultQueryPlanner().physical_plan_limit(scan, 7, 4, sort_exprs=[(col("f64", 3), False), (col("s", 4), True)])
    assert isinstance(plan, q.Limit) and isinstance(plan.input, q.Sort) and plan.input.limit == 11
    _batches_equal(plan.execute(), qoracle.execute(plan))
    # sorting a join's (deferred-gather) output
    on = [(col("u8", 7), col("u8", 7))]
    join = q.HashJoinExec.try_new(table_scan(schema, [batch.slice(0, 40)]), table_scan(schema, [batch.slice(40, 60)]), q.JoinType.Inner, on, None)
    plan = q.Sort([q.PhysicalSortExpr(col("i32", 9), q.SortOptions(True, False)), q.PhysicalSortExpr(col("row", 8), q.SortOptions(False, True))], join)
    _batches_equal(plan.execute(), qoracle.execute(plan))
