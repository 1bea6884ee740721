"""The CPU oracle against Arrow C++ (pyarrow / Acero), an implementation independent of both the reference's arrow-rs and
this repo, on the generic operators where their semantics coincide (SURVEY §8c "independent cross-check available here"):
filter with NULL predicates, multi-key sort with NULLs first, integer GROUP BY aggregates, equi-joins of every type as row
multisets. Decimal precision rules and AVG are NOT checked here — Arrow C++ differs from arrow-rs there and the
reference's own vectors pin them (tests/test_reference_goldens.py)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator

from .helpers import col, lit_i64, rows_of, table_scan

I64 = pa.int64()


def _key(row):
    return tuple((v is not None, v if v is not None else 0) for v in row)


def _table(rng, n, nkeys, names, null_frac=0.1):
    arrays = [pa.array(rng.integers(0, nkeys, n), type=I64, mask=rng.random(n) < null_frac)]
    for _ in names[1:]:
        arrays.append(pa.array(rng.integers(-1000, 1000, n), type=I64, mask=rng.random(n) < null_frac))
    schema = pa.schema([pa.field(nm, I64) for nm in names])
    return schema, pa.RecordBatch.from_arrays(arrays, schema=schema)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_filter_drops_null_predicates_like_arrow_cpp(oracle, seed):
    rng = np.random.default_rng(seed)
    schema, batch = _table(rng, 5000, 50, ["k", "v"])
    pred = q.BinaryExpr(q.BinaryExpr(col("k", 0), Operator.Lt, lit_i64(20)), Operator.Or, q.BinaryExpr(col("v", 1), Operator.Gt, lit_i64(500)))
    got = rows_of(oracle.execute(q.Filter(table_scan(schema, [batch.slice(0, 1234), batch.slice(1234)]), pred)))
    mask = pc.or_kleene(pc.less(batch.column(0), 20), pc.greater(batch.column(1), 500))
    want = rows_of([batch.filter(mask, null_selection_behavior="drop")])
    assert got == want and 0 < len(got) < 5000


@pytest.mark.parametrize("seed", [4, 5])
def test_sort_is_arrow_cpps_stable_sort_with_nulls_first(oracle, seed):
    rng = np.random.default_rng(seed)
    schema, batch = _table(rng, 4000, 12, ["a", "b", "c"], null_frac=0.15)
    plan = q.Sort([q.PhysicalSortExpr(col("a", 0), q.SortOptions(descending=True, nulls_first=True)),
                   q.PhysicalSortExpr(col("b", 1), q.SortOptions(descending=False, nulls_first=True))],
                  table_scan(schema, [batch.slice(0, 1500), batch.slice(1500)]))
    got = rows_of(oracle.execute(plan))
    idx = pc.sort_indices(pa.Table.from_batches([batch]), sort_keys=[("a", "descending"), ("b", "ascending")], null_placement="at_start")
    assert got == rows_of([batch.take(idx)])


@pytest.mark.parametrize("seed", [6, 7])
def test_integer_group_by_matches_acero(oracle, seed):
    rng = np.random.default_rng(seed)
    schema, batch = _table(rng, 6000, 40, ["k", "v"], null_frac=0.2)
    aggs = [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(col("v", 1)), q.CountAggregateExpr(lit_i64(1)),
            q.MinAggregateExpr(col("v", 1), I64), q.MaxAggregateExpr(col("v", 1), I64)]
    got = sorted(rows_of(oracle.execute(q.HashAggregate(None, table_scan(schema, [batch]), [col("k", 0)], aggs))), key=_key)
    t = pa.Table.from_batches([batch]).group_by("k").aggregate([("v", "sum"), ("v", "count"), ([], "count_all"), ("v", "min"), ("v", "max")])
    want = sorted(zip(t.column("k").to_pylist(), t.column("v_sum").to_pylist(), t.column("v_count").to_pylist(),
                      t.column("count_all").to_pylist(), t.column("v_min").to_pylist(), t.column("v_max").to_pylist()), key=_key)
    assert got == [tuple(r) for r in want] and any(r[0] is None for r in got)      # NULL keys form one group in both


ACERO = {JoinType.Inner: "inner", JoinType.Left: "left outer", JoinType.Right: "right outer", JoinType.Full: "full outer",
         JoinType.LeftSemi: "left semi", JoinType.LeftAnti: "left anti"}


@pytest.mark.parametrize("jt", list(JoinType))
def test_equi_join_row_multisets_match_acero(oracle, jt):
    rng = np.random.default_rng(11 + int(jt))
    ls, lb = _table(rng, 900, 60, ["lk", "lv"])
    rs, rb = _table(rng, 1400, 60, ["rk", "rv"])
    plan = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb.slice(0, 500), rb.slice(500)]), jt, [(col("lk", 0), col("rk", 0))], None)
    got = sorted(rows_of(oracle.execute(plan)), key=_key)
    t = pa.Table.from_batches([lb]).join(pa.Table.from_batches([rb]), keys="lk", right_keys="rk", join_type=ACERO[jt], coalesce_keys=False)
    names = ["lk", "lv"] if jt in (JoinType.LeftSemi, JoinType.LeftAnti) else ["lk", "lv", "rk", "rv"]
    want = sorted(zip(*[t.column(nm).to_pylist() for nm in names]), key=_key)
    assert got == [tuple(r) for r in want] and len(got) > 50       # NULL keys never match in either
