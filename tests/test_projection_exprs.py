"""Projection + CASE / LIKE (physical/plan/projection.rs, physical/expr/{case,like}.rs). The reference
has no unit vectors for these (only TPC-H SF0.01 outputs whose inputs are not available offline): the oracle restates
arrow's `zip` and `like`, and is cross-checked here against Arrow C++ (pyarrow.compute.if_else / match_like) — an
independent implementation of the same kernels. The gpu tests compare the HIP path with the oracle."""
from __future__ import annotations

import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import qurious_amd as q
from oracle import qoracle
from qurious_amd import Operator
from qurious_amd import ScalarValue as S

from .helpers import col, rows_of, table_scan

D = decimal.Decimal
DEC = pa.decimal128(15, 2)
STRINGS = ["", "a", "ab", "abc", "PROMO BRUSHED", "STANDARD POLISHED TIN", "forest green", "fo%rest", "a_c", "a\\c", "é", "éa", "naïve café",
           "greenish", "evergreen", "x" * 40, "%", "_", "日本語テキスト", "MAIL", "SHIP", "special requests", "no special needs requests"]
PATTERNS = ["%", "", "a", "a%", "%a", "%a%", "_", "__", "a_c", "%green%", "forest%", "%BRUSHED", "PROMO%", "%special%requests%", "fo\\%rest", "a\\_c",
            "a\\\\c", "_a", "é%", "%é", "%_", "_%_", "%日本%", "___", "x%x", "%%", "%_%_%"]


def _table(rng, n):
    m = lambda p=0.1: rng.random(n) < p   # noqa: E731
    cols = {
        "i": pa.array(rng.integers(-20, 20, n), type=pa.int64(), mask=m()),
        "j": pa.array(rng.integers(-20, 20, n), type=pa.int64(), mask=m()),
        "d": pa.array([D(int(v)).scaleb(-2) for v in rng.integers(-10**6, 10**6, n)], type=DEC, mask=m()),
        "e": pa.array([D(int(v)).scaleb(-2) for v in rng.integers(1, 10**4, n)], type=DEC),
        "f": pa.array(rng.normal(size=n), type=pa.float64(), mask=m()),
        "s": pa.array([STRINGS[v] for v in rng.integers(0, len(STRINGS), n)], type=pa.string(), mask=m()),
        "b": pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=m()),
        "day": pa.array(rng.integers(9000, 9100, n), type=pa.int32()).cast(pa.date32()),
    }
    schema = pa.schema([pa.field(k, v.type, True) for k, v in cols.items()])
    return schema, pa.RecordBatch.from_arrays(list(cols.values()), schema=schema)


def _like(pattern, negated=False):
    return q.Like(negated, col("s", 5), q.Literal(S.Utf8(pattern)))


def test_oracle_like_matches_arrow_cpp():
    arr = pa.array(STRINGS + [None], type=pa.string())
    batch = pa.RecordBatch.from_arrays([arr] * 6, names=["a", "b", "c", "d", "e", "s"])
    for pat in PATTERNS:
        want = pc.match_like(arr, pat)
        got = qoracle.evaluate(_like(pat), batch)
        assert got.to_pylist() == want.to_pylist(), pat
        got_n = qoracle.evaluate(_like(pat, True), batch)
        assert got_n.to_pylist() == pc.invert(want).to_pylist(), pat


def _pattern_table(rng, n):
    """strings next to a PATTERN COLUMN: every (string, pattern) pair of the two lists occurs, NULLs on both sides"""
    si, pi = rng.integers(0, len(STRINGS), n), rng.integers(0, len(PATTERNS), n)
    si[:len(STRINGS) * len(PATTERNS)] = np.repeat(np.arange(len(STRINGS)), len(PATTERNS))[:n]
    pi[:len(STRINGS) * len(PATTERNS)] = np.tile(np.arange(len(PATTERNS)), len(STRINGS))[:n]
    schema = pa.schema([pa.field("s", pa.string()), pa.field("p", pa.string()), pa.field("k", pa.int64())])
    batch = pa.RecordBatch.from_arrays([pa.array([STRINGS[v] for v in si], type=pa.string(), mask=rng.random(n) < 0.05),
                                        pa.array([PATTERNS[v] for v in pi], type=pa.string(), mask=rng.random(n) < 0.05),
                                        pa.array(np.arange(n), type=pa.int64())], schema=schema)
    return schema, batch


def test_oracle_like_with_a_pattern_column_matches_arrow_cpp_row_by_row():
    """like.rs:28-43 evaluates the pattern expression per batch and hands arrow's `like` two ARRAYS: row i is matched against
    pattern i. Arrow C++ has no array-pattern kernel; it is asked once per distinct pattern and the rows are picked."""
    rng = np.random.default_rng(9)
    schema, batch = _pattern_table(rng, len(STRINGS) * len(PATTERNS) + 500)
    for neg in (False, True):
        got = qoracle.evaluate(q.Like(neg, col("s", 0), col("p", 1)), batch).to_pylist()
        strings, patterns = batch.column(0), batch.column(1).to_pylist()
        per_pattern = {pat: pc.match_like(strings, pat).to_pylist() for pat in set(patterns) if pat is not None}
        want = [None if (pat is None or per_pattern[pat][i] is None) else (per_pattern[pat][i] != neg) for i, pat in enumerate(patterns)]
        assert got == want


def test_oracle_case_matches_arrow_cpp():
    rng = np.random.default_rng(1)
    schema, batch = _table(rng, 500)
    cond1 = q.BinaryExpr(col("i", 0), Operator.Gt, q.Literal(S.Int64(3)))
    cond2 = q.BinaryExpr(col("j", 1), Operator.Lt, q.Literal(S.Int64(0)))
    expr = q.CaseExpr([(cond1, col("d", 2)), (cond2, col("e", 3))], q.CastExpr(q.Literal(S.Int64(0)), DEC))
    got = qoracle.evaluate(expr, batch)
    c1 = pc.fill_null(pc.greater(batch.column(0), 3), False)
    c2 = pc.fill_null(pc.less(batch.column(1), 0), False)
    want = pc.if_else(c1, batch.column(2), pc.if_else(c2, batch.column(3), pa.scalar(D("0.00"), type=DEC)))
    assert got.type == DEC and got.to_pylist() == want.to_pylist()
    with pytest.raises(qoracle.OracleError, match="same data type"):
        qoracle.evaluate(q.CaseExpr([(cond1, col("d", 2))], col("i", 0)), batch)
    with pytest.raises(qoracle.OracleError, match="must be boolean"):
        qoracle.evaluate(q.CaseExpr([(col("i", 0), col("d", 2))], col("e", 3)), batch)


def _plans(schema, scan):
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))
    promo = q.CaseExpr([(_like("PROMO%"), q.BinaryExpr(col("d", 2), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("e", 3))))],
                       q.CastExpr(q.Literal(S.Int64(0)), pa.decimal128(38, 4)))                       # Q14's numerator
    high = q.CaseExpr([(q.BinaryExpr(q.BinaryExpr(col("s", 5), Operator.Eq, q.Literal(S.Utf8("MAIL"))), Operator.Or,
                                     q.BinaryExpr(col("s", 5), Operator.Eq, q.Literal(S.Utf8("SHIP")))), q.Literal(S.Int64(1)))],
                      q.Literal(S.Int64(0)))                                                          # Q12's high_line_count
    exprs = [
        ("i", col("i", 0)),                                                                            # shared column
        ("i_plus_j", q.BinaryExpr(col("i", 0), Operator.Add, col("j", 1))),
        ("ratio", q.BinaryExpr(col("d", 2), Operator.Div, col("e", 3))),                               # decimal / decimal -> Float64
        ("promo", promo), ("high", high),
        ("neg_f", q.Negative(col("f", 4))),
        ("s_is_null", q.IsNull(col("s", 5))),
        ("like_green", _like("%green%")), ("not_like", _like("_a%", True)),
        ("b_or", q.BinaryExpr(col("b", 6), Operator.Or, q.BinaryExpr(col("i", 0), Operator.Gt, col("j", 1)))),
        ("day", col("day", 7)),
        ("case_date", q.CaseExpr([(col("b", 6), col("day", 7))], q.CastExpr(q.Literal(S.Utf8("1995-01-01")), pa.date32()))),
        ("case_f", q.CaseExpr([(q.IsNull(col("f", 4)), q.Literal(S.Float64(-1.0))), (col("b", 6), col("f", 4))], q.Negative(col("f", 4)))),
        ("lit", q.Literal(S.Int32(7))),
    ]
    return q.Projection(None, scan, [e for _, e in exprs])


def test_oracle_projection_shapes():
    rng = np.random.default_rng(2)
    schema, batch = _table(rng, 300)
    scan = table_scan(schema, [batch.slice(0, 100), batch.slice(100, 0), batch.slice(100, 200)])
    out = qoracle.execute(_plans(schema, scan))
    assert [b.num_rows for b in out] == [100, 0, 200] and out[0].num_columns == 14
    assert out[0].schema.field(2).type == pa.float64() and out[0].schema.field(3).type == pa.decimal128(38, 4)


# ------------------------------------------------------------------------------------------------------------ gpu
def _same(got, want):
    assert [b.num_rows for b in got] == [b.num_rows for b in want]
    assert [str(f.type) for f in got[0].schema] == [str(f.type) for f in want[0].schema] if got else True
    for g, w in zip(got, want):
        for k in range(g.num_columns):
            gv, wv = g.column(k).to_pylist(), w.column(k).to_pylist()
            if pa.types.is_floating(g.column(k).type):
                assert len(gv) == len(wv)
                for a, b in zip(gv, wv):
                    assert (a is None) == (b is None) and (a is None or a == b or (a != a and b != b) or abs(a - b) <= 1e-12 * abs(b)), (k, a, b)
            else:
                assert gv == wv, g.schema.field(k).name


@pytest.mark.gpu
def test_gpu_projection_case_like_vs_oracle():
    q.get_context()
    rng = np.random.default_rng(3)
    schema, batch = _table(rng, 20_000)
    cuts = [0, 5000, 5000, 12_345, 20_000]
    scan = table_scan(schema, [batch.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    plan = _plans(schema, scan)
    _same(plan.execute(), qoracle.execute(plan))
    # every pattern, as a Filter predicate and as a projected column
    for pat in PATTERNS:
        for neg in (False, True):
            f = q.Filter(scan, _like(pat, neg))
            assert rows_of(f.execute()) == rows_of(qoracle.execute(f)), (pat, neg)
    # empty input, zero batches
    for src in (table_scan(schema, [batch.slice(0, 0)]), q.Scan(schema, q.MemoryTable.try_new(schema, []))):
        p2 = _plans(schema, src)
        _same(p2.execute(), qoracle.execute(p2))


@pytest.mark.gpu
def test_gpu_like_with_a_pattern_column_vs_oracle():
    """Round 4: LIKE / NOT LIKE whose pattern is a COLUMN (VERDICT r03 missing #4): as a Filter predicate, as a projected Boolean,
    fused into an aggregate's scan filter and inside CASE; a computed pattern stays unsupported."""
    q.get_context()
    rng = np.random.default_rng(10)
    schema, batch = _pattern_table(rng, 30_000)
    scan = table_scan(schema, [batch.slice(0, 11_111), batch.slice(11_111)])
    for neg in (False, True):
        like = q.Like(neg, col("s", 0), col("p", 1))
        f = q.Filter(scan, like)
        assert rows_of(f.execute()) == rows_of(qoracle.execute(f)), neg
        proj = q.Projection(None, scan, [col("k", 2), like, q.CaseExpr([(like, col("k", 2))], q.Literal(S.Int64(-1)))])
        assert rows_of(proj.execute()) == rows_of(qoracle.execute(proj)), neg
        agg = q.HashAggregate(None, table_scan(schema, [batch], like), [col("p", 1)], [q.CountAggregateExpr(q.Literal(S.Int64(1))), q.SumAggregateExpr(col("k", 2), pa.int64())])
        assert sorted(rows_of(agg.execute()), key=repr) == sorted(rows_of(qoracle.execute(agg)), key=repr), neg
    computed = q.Like(False, col("s", 0), q.CaseExpr([(q.IsNull(col("k", 2)), col("p", 1))], col("s", 0)))
    with pytest.raises(q.QuriousError):
        q.Filter(scan, computed).execute()


def _string_cases():
    """CASE expressions that PRODUCE strings: literals, columns, NULL branches, nesting"""
    day = q.CastExpr(q.Literal(S.Int32(9050)), pa.date32())
    lit = lambda v: q.Literal(S.Utf8(v))   # noqa: E731
    c1 = q.CaseExpr([(q.BinaryExpr(col("i", 0), Operator.Eq, q.Literal(S.Int64(1))), lit("a"))], lit("b"))                 # type.slt:51's shape
    c2 = q.CaseExpr([(q.BinaryExpr(col("day", 7), Operator.Lt, day), col("s", 5)), (q.BinaryExpr(col("i", 0), Operator.Gt, col("j", 1)), lit("i above j"))], lit(""))
    c3 = q.CaseExpr([(col("b", 6), q.CaseExpr([(q.IsNull(col("s", 5)), lit("<null>"))], col("s", 5)))], q.Literal(S.Utf8(None)))
    c4 = q.CaseExpr([(q.Like(False, col("s", 5), lit("%green%")), lit("a rather long constant that is longer than any column value, 日本語"))], col("s", 5))
    return [c1, c2, c3, c4, lit("tag")]


def test_oracle_string_case_of_the_reference_slt():
    """tests/sql/type.slt:42-56 (SimpleCaseExpr) — the one vector the reference holds for a CASE: t_case(x int not null) = (1), (2);
    select case x when 1 then 'a' else 'b' end -> a, b (rowsort). The planner lowers `case x when 1` to `x = 1` (case.rs)."""
    schema = pa.schema([pa.field("x", pa.int32(), False)])
    batch = pa.RecordBatch.from_arrays([pa.array([1, 2], type=pa.int32())], schema=schema)
    case = q.CaseExpr([(q.BinaryExpr(col("x", 0), Operator.Eq, q.Literal(S.Int32(1))), q.Literal(S.Utf8("a")))], q.Literal(S.Utf8("b")))
    plan = q.Projection(None, table_scan(schema, [batch]), [case])
    assert sorted(r[0] for r in rows_of(qoracle.execute(plan))) == ["a", "b"]


@pytest.mark.gpu
def test_gpu_projection_that_produces_strings():
    """Round 4 (VERDICT r03 missing #4): a projected expression of Utf8 type — two passes of the generated kernel (lengths, then
    bytes at the scanned offsets). The reference's own vector (type.slt:51) through the HIP path, then random tables: CASE over
    literals and columns, NULL branches, nesting, a LIKE condition, a literal column; multi-batch, empty and zero-batch inputs;
    and a consumer above it (GROUP BY the computed string)."""
    q.get_context()
    schema1 = pa.schema([pa.field("x", pa.int32(), False)])
    batch1 = pa.RecordBatch.from_arrays([pa.array([1, 2], type=pa.int32())], schema=schema1)
    case = q.CaseExpr([(q.BinaryExpr(col("x", 0), Operator.Eq, q.Literal(S.Int32(1))), q.Literal(S.Utf8("a")))], q.Literal(S.Utf8("b")))
    got = q.Projection(None, table_scan(schema1, [batch1]), [case]).execute()
    assert sorted(r[0] for r in rows_of(got)) == ["a", "b"] and got[0].schema.field(0).type == pa.string()
    rng = np.random.default_rng(12)
    schema, batch = _table(rng, 25_000)
    cuts = [0, 4000, 4000, 13_001, 25_000]
    scan = table_scan(schema, [batch.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    plan = q.Projection(None, scan, [col("i", 0)] + _string_cases() + [col("s", 5)])
    _same(plan.execute(), qoracle.execute(plan))
    for src in (table_scan(schema, [batch.slice(0, 0)]), q.Scan(schema, q.MemoryTable.try_new(schema, []))):
        p2 = q.Projection(None, src, _string_cases())
        _same(p2.execute(), qoracle.execute(p2))
    grouped = q.HashAggregate(None, q.Projection(None, scan, [_string_cases()[1], col("i", 0)]), [col("c", 0)],
                              [q.CountAggregateExpr(q.Literal(S.Int64(1))), q.SumAggregateExpr(col("i", 1), pa.int64())])
    assert sorted(rows_of(grouped.execute()), key=repr) == sorted(rows_of(qoracle.execute(grouped)), key=repr)
    flt = q.Filter(q.Projection(None, scan, [_string_cases()[3], col("i", 0)]), q.Like(False, col("c", 0), q.Literal(S.Utf8("%constant%"))))
    assert rows_of(flt.execute()) == rows_of(qoracle.execute(flt))


@pytest.mark.gpu
def test_gpu_case_inside_aggregate_and_errors():
    """Q12 / Q14 shapes: SUM(CASE WHEN ... THEN x ELSE 0 END) fused into the aggregation kernel"""
    q.get_context()
    rng = np.random.default_rng(4)
    schema, batch = _table(rng, 30_000)
    scan = table_scan(schema, [batch])
    proj = _plans(schema, scan)
    promo, high = proj.exprs[3], proj.exprs[4]
    agg = q.HashAggregate(None, scan, [col("day", 7)], [q.SumAggregateExpr(promo, pa.decimal128(38, 4)), q.SumAggregateExpr(high, pa.int64()),
                                                        q.CountAggregateExpr(q.Literal(S.Int64(1)))])
    assert sorted(rows_of(agg.execute())) == sorted(rows_of(qoracle.execute(agg)))
    with pytest.raises(q.QuriousError, match="same data type"):
        q.Projection(None, scan, [q.CaseExpr([(col("b", 6), col("d", 2))], col("i", 0))]).execute()
    with pytest.raises(q.QuriousError, match="must be boolean"):
        q.Projection(None, scan, [q.CaseExpr([(col("i", 0), col("d", 2))], col("e", 3))]).execute()
    # both CASE branches are evaluated for every row, like the reference's full-array evaluation: a division by zero in
    # the branch that is never selected still fails the query
    zero = q.BinaryExpr(col("i", 0), Operator.Div, q.BinaryExpr(col("i", 0), Operator.Sub, col("i", 0)))
    guarded = q.CaseExpr([(q.Literal(S.Boolean(False)), zero)], col("i", 0))
    with pytest.raises(q.QuriousError, match="Divide by zero"):
        q.Projection(None, scan, [guarded]).execute()
    with pytest.raises(qoracle.OracleError, match="Divide by zero"):
        qoracle.execute(q.Projection(None, scan, [guarded]))
