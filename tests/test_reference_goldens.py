"""Every golden vector the reference's own tests hold for the hot path (SURVEY Appendix B,
tests/golden/reference_vectors.json), replayed through an `engine` (tests/conftest.py): the CPU oracle
(oracle/qoracle.c — this pins it; runs under -m "not gpu") and the HIP path through the C ABI (`[hip]`, marked gpu).
The SipHash / JoinHashMap vectors concern structures only the oracle has (the HIP path keys its tables by the real key
words) and stay oracle-only."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator
from qurious_amd import ScalarValue as S

from .helpers import build_table_scan_i32, col, lit_i64, rows_of, sorted_rows, table_scan


# ---------------------------------------------------------------- SipHash-1-3 (independent pure-Python restatement)
def _siphash13_py(data: bytes, k0=0, k1=0) -> int:
    M = (1 << 64) - 1
    v0, v1, v2, v3 = k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573
    rotl = lambda x, b: ((x << b) | (x >> (64 - b))) & M

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & M; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32)
        v2 = (v2 + v3) & M; v3 = rotl(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & M; v3 = rotl(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & M; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(data)
    for i in range(0, n - n % 8, 8):
        m = int.from_bytes(data[i:i + 8], "little")
        v3 ^= m
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    b = ((n & 0xff) << 56) | int.from_bytes(data[n - n % 8:], "little")
    v3 ^= b
    v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= b
    v2 ^= 0xff
    for _ in range(3):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


def test_siphash13_matches_independent_restatement(oracle):
    rng = np.random.default_rng(7)
    for n in list(range(0, 40)) + [63, 64, 65, 255]:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle.siphash13(data) == _siphash13_py(data)
    # streaming writes (how create_hashes feeds multi-column keys) == one-shot over the concatenation
    parts = [b"abc", b"\xff", (12345).to_bytes(8, "little"), b"", (7).to_bytes(4, "little")]
    assert oracle.siphash13_chunks(parts) == _siphash13_py(b"".join(parts))


def test_create_hashes_byte_stream(oracle):
    """utils/array.rs:171-210: ints as LE bytes, Utf8 as bytes + 0xFF, NULL feeds nothing, columns chained."""
    i64 = pa.array([1, None, -5], type=pa.int64())
    s = pa.array(["A", "xy", None], type=pa.string())
    d = pa.array([decimal.Decimal("1.25"), decimal.Decimal("-3.00"), None], type=pa.decimal128(15, 2))
    dt = pa.array([10470, 0, 3], type=pa.int32()).cast(pa.date32())
    h = oracle.create_hashes([i64, s, d, dt])
    exp0 = _siphash13_py((1).to_bytes(8, "little", signed=True) + b"A\xff" + (125).to_bytes(16, "little", signed=True) + (10470).to_bytes(4, "little"))
    exp1 = _siphash13_py(b"xy\xff" + (-300).to_bytes(16, "little", signed=True) + (0).to_bytes(4, "little"))
    exp2 = _siphash13_py((-5).to_bytes(8, "little", signed=True) + (3).to_bytes(4, "little"))
    assert [int(x) for x in h] == [exp0, exp1, exp2]
    with pytest.raises(oracle.OracleError, match="Unsupported data type in hasher"):
        oracle.create_hashes([pa.array([1.0])])


# ---------------------------------------------------------------- JoinHashMap vectors (hash_join.rs:700-886)
def test_join_hash_map_update_vectors(oracle, golden):
    for case in golden["join_hash_map_update"]["cases"]:
        hashes = np.array(case["hashes"], dtype=np.uint64)
        m = oracle.JoinHashMap(len(hashes))
        m.update(hashes, list(range(len(hashes))), case["delete_offset"])
        for k, v in case["map"].items():
            assert m.get(int(k)) == v
        assert m.map_len() == len(case["map"])
        assert m.next() == case["next"]
        if case["is_distinct"] is not None:
            assert m.is_distinct() == case["is_distinct"]


def test_join_hash_map_get_matches_vectors(oracle, golden):
    for case in golden["join_hash_map_matches"]["cases"]:
        hashes = np.array(case["build"], dtype=np.uint64)
        m = oracle.JoinHashMap(len(hashes))
        m.update(hashes, list(range(len(hashes))), 0)
        ii, mi = m.get_matches_indices(case["probe"])
        assert ii == case["input_indices"]
        assert mi == case["match_indices"]


# ---------------------------------------------------------------- HashJoinExec ordered outputs (hash_join.rs:396-698, 889-914)
def _join_case(case):
    left = build_table_scan_i32(case["left"])
    right = build_table_scan_i32(case["right"])
    lnames, rnames = list(case["left"]), list(case["right"])
    on = [(col(lnames[l], l), col(rnames[r], r)) for l, r in case["on"]]
    return q.HashJoinExec.try_new(left, right, JoinType[case["join_type"]], on, None)


def test_hash_join_exec_goldens(engine, golden):
    for case in golden["hash_join_exec"]["cases"]:
        got = rows_of(engine.execute(_join_case(case)))
        assert got == [tuple(r) for r in case["expected"]], case["name"]


# ---------------------------------------------------------------- expression vectors (binary.rs:100-251)
def test_binary_expr_int32_vectors(engine, golden):
    for case in golden["binary_expr_int32"]["cases"]:
        batch = pa.RecordBatch.from_arrays([pa.array(case["l"], type=pa.int32()), pa.array(case["r"], type=pa.int32())], names=["a", "b"])
        e = q.BinaryExpr(col("a", 0), Operator[case["op"]], col("b", 1))
        assert engine.evaluate(e, batch).to_pylist() == case["expected"], case["op"]


def test_binary_expr_boolean_vectors(engine, golden):
    g = golden["binary_expr_boolean"]
    batch = pa.RecordBatch.from_arrays([pa.array(g["l"]), pa.array(g["r"])], names=["a", "b"])
    for op in ("And", "Or"):
        assert engine.evaluate(q.BinaryExpr(col("a", 0), Operator[op], col("b", 1)), batch).to_pylist() == g[op]


def test_kleene_logic(engine):
    a = pa.array([True, True, True, False, False, False, None, None, None])
    b = pa.array([True, False, None, True, False, None, True, False, None])
    batch = pa.RecordBatch.from_arrays([a, b], names=["a", "b"])
    assert engine.evaluate(q.BinaryExpr(col("a", 0), Operator.And, col("b", 1)), batch).to_pylist() == \
        [True, False, None, False, False, False, None, False, None]
    assert engine.evaluate(q.BinaryExpr(col("a", 0), Operator.Or, col("b", 1)), batch).to_pylist() == \
        [True, True, True, True, False, None, True, None, None]


def test_binary_expr_decimal_vector(engine, golden):
    g = golden["binary_expr_decimal"]
    t = pa.decimal128(*g["input_type"])
    dec = lambda u: decimal.Decimal(u).scaleb(-g["input_type"][1])
    batch = pa.RecordBatch.from_arrays([pa.array([dec(g["l_extendedprice"])], type=t), pa.array([dec(g["l_discount"])], type=t)],
                                       names=["l_extendedprice", "l_discount"])
    one = q.CastExpr(q.Literal(S.Int16(1)), t)
    e = q.BinaryExpr(col("l_extendedprice", 0), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("l_discount", 1)))
    out = engine.evaluate(e, batch)
    assert out.type == pa.decimal128(*g["expected_type"])
    assert out[0].as_py() == decimal.Decimal(g["expected_unscaled"]).scaleb(-g["expected_type"][1])


def test_q1_type_rules(engine):
    """1 - l_discount -> (23,2); price * that -> (38,4); * (1 + l_tax) -> (38,6) (q1.slt:24 scale digits)."""
    t = pa.decimal128(15, 2)
    D = decimal.Decimal
    batch = pa.RecordBatch.from_arrays([pa.array([D("100.00")], type=t), pa.array([D("0.05")], type=t), pa.array([D("0.08")], type=t)],
                                       names=["p", "d", "x"])
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))
    e1 = q.BinaryExpr(one, Operator.Sub, col("d", 1))
    e2 = q.BinaryExpr(col("p", 0), Operator.Mul, e1)
    e3 = q.BinaryExpr(e2, Operator.Mul, q.BinaryExpr(one, Operator.Add, col("x", 2)))
    assert engine.evaluate(e1, batch).type == pa.decimal128(23, 2)
    assert engine.evaluate(e2, batch).type == pa.decimal128(38, 4)
    r = engine.evaluate(e3, batch)
    assert r.type == pa.decimal128(38, 6)
    assert r[0].as_py() == D("102.600000")


# ---------------------------------------------------------------- .slt goldens restated as physical plans
I64 = pa.int64()


def _t(names, rows, types=None):
    types = types or [I64] * len(names)
    return table_scan(pa.schema([pa.field(n, t, True) for n, t in zip(names, types)]), [tuple(r) for r in rows])


def _filter(scan, pred):
    return q.Filter(scan, pred)


def _agg(inp, groups, aggs):
    cls = q.HashAggregate if groups else None
    names = [f"k{i}" for i in range(len(groups))] + [f"a{i}" for i in range(len(aggs))]
    types = [I64] * len(names)
    schema = pa.schema([pa.field(n, t, True) for n, t in zip(names, types)])
    return q.HashAggregate(schema, inp, groups, aggs) if groups else q.NoGroupingAggregate(schema, inp, aggs)


def _sum(e):
    return q.SumAggregateExpr(e, I64)


def test_slt_where(engine, golden):
    g = golden["slt"]["where_t1"]
    t = lambda: _t(["v1", "v2"], g["rows"])
    v1, v2 = col("v1", 0), col("v2", 1)
    assert rows_of(engine.execute(_filter(t(), q.BinaryExpr(v1, Operator.Gt, v2)))) == [tuple(r) for r in g["v1_gt_v2"]]
    assert rows_of(engine.execute(_filter(t(), q.BinaryExpr(v2, Operator.Gt, lit_i64(2))))) == [tuple(r) for r in g["v2_gt_2"]]
    pred = q.BinaryExpr(q.BinaryExpr(v1, Operator.Eq, lit_i64(1)), Operator.Or, q.BinaryExpr(v2, Operator.Eq, lit_i64(2)))
    assert rows_of(engine.execute(_filter(t(), pred))) == [tuple(r) for r in g["v1_eq_1_or_v2_eq_2"]]
    plan = _agg(_filter(t(), q.BinaryExpr(v1, Operator.NotEq, lit_i64(1))), [], [_sum(v2)])
    assert rows_of(engine.execute(plan)) == [(g["sum_v2_where_v1_ne_1"],)]
    g2 = golden["slt"]["where_t2"]
    for op, key in ((Operator.Lt, "sum_v2_v1_lt_1"), (Operator.LtEq, "sum_v2_v1_le_1"), (Operator.GtEq, "sum_v2_v1_ge_1")):
        plan = _agg(_filter(_t(["v1", "v2"], g2["rows"]), q.BinaryExpr(v1, op, lit_i64(1))), [], [_sum(v2)])
        assert rows_of(engine.execute(plan)) == [(g2[key],)]
    g3 = golden["slt"]["where_t3"]
    out = rows_of(engine.execute(_filter(_t(["v1", "v2"], g3["rows"]), q.IsNull(v1))))
    assert [r[1] for r in out] == g3["v2_where_v1_is_null"]
    out = rows_of(engine.execute(_filter(_t(["v1", "v2"], g3["rows"]), q.IsNotNull(v1))))
    assert [r[1] for r in out] == g3["v2_where_v1_is_not_null"]


def test_slt_filter_null(engine, golden):
    g = golden["slt"]["filter_null"]
    plan = _filter(_t(["v1", "v2"], g["rows"]), q.BinaryExpr(col("v1", 0), Operator.Gt, lit_i64(1)))
    assert rows_of(engine.execute(plan)) == [tuple(r) for r in g["v1_gt_1"]]


def test_slt_aggregation(engine, golden):
    g = golden["slt"]["aggregation"]
    types = [I64, I64, pa.float64()]
    t = lambda rows=g["rows"]: _t(["v1", "v2", "v3"], rows, types)
    v1, v2, v3 = col("v1", 0), col("v2", 1), col("v3", 2)
    out = rows_of(engine.execute(q.NoGroupingAggregate(None, t(), [_sum(v1), q.SumAggregateExpr(v3, pa.float64())])))
    assert out[0][0] == g["sum_v1"] and abs(out[0][1] - g["sum_v3"]) < 1e-9
    out = rows_of(engine.execute(q.NoGroupingAggregate(None, t(), [q.MinAggregateExpr(v1, I64), q.MaxAggregateExpr(v1, I64), q.CountAggregateExpr(lit_i64(1))])))
    assert out == [(g["min_v1"], g["max_v1"], g["count"])]
    plan = q.NoGroupingAggregate(None, _filter(t(), q.BinaryExpr(v2, Operator.Gt, lit_i64(3))), [q.MaxAggregateExpr(v1, I64)])
    assert rows_of(engine.execute(plan)) == [(g["max_v1_where_v2_gt_3"],)]
    out = sorted_rows(engine.execute(q.HashAggregate(None, t(), [v2], [_sum(v1)])))
    assert out == sorted((r[1], r[0]) for r in g["sum_v1_group_by_v2"])
    # empty table: zero batches
    empty = q.Scan(pa.schema([pa.field("v1", I64), pa.field("v2", I64)]), q.MemoryTable.try_new(pa.schema([pa.field("v1", I64), pa.field("v2", I64)]), []))
    assert rows_of(engine.execute(q.NoGroupingAggregate(None, empty, [q.CountAggregateExpr(lit_i64(1)), _sum(col("v1", 0))]))) == [(g["empty_count"], g["empty_sum"])]
    assert engine.execute(q.HashAggregate(None, empty, [col("v1", 0)], [q.CountAggregateExpr(lit_i64(1))])) == []


def test_slt_group_by_and_having(engine, golden):
    g = golden["slt"]["group_by"]
    t = _t(["v1", "v2"], g["rows"])
    key = q.BinaryExpr(col("v2", 1), Operator.Add, lit_i64(1))
    out = sorted_rows(engine.execute(q.HashAggregate(None, t, [key], [_sum(col("v1", 0))])))
    assert out == sorted(tuple(r) for r in g["v2_plus_1__sum_v1"])
    out = sorted_rows(engine.execute(q.HashAggregate(None, t, [key], [_sum(col("v1", 0)), q.CountAggregateExpr(lit_i64(1))])))
    assert out == sorted((r[1], r[0], r[2]) for r in g["sum_v1__v2_plus_1__count"])
    h = golden["slt"]["having"]
    t = _t(["x", "y"], h["rows"])
    agg = q.HashAggregate(None, t, [col("y", 1)], [_sum(col("x", 0))])
    out = rows_of(engine.execute(_filter(agg, q.BinaryExpr(col("k0", 0), Operator.Eq, lit_i64(2)))))
    assert out == [tuple(r) for r in h["y_sumx_having_y_eq_2"]]
    agg = q.HashAggregate(None, t, [col("y", 1)], [q.CountAggregateExpr(col("x", 0))])
    out = rows_of(engine.execute(_filter(agg, q.BinaryExpr(col("a0", 1), Operator.Gt, lit_i64(1)))))
    assert [(r[1], r[0]) for r in out] == [tuple(r) for r in h["countx_y_having_gt_1"]]
    agg = q.HashAggregate(None, t, [col("x", 0)], [q.MaxAggregateExpr(col("y", 1), I64)])
    out = rows_of(engine.execute(_filter(agg, q.BinaryExpr(col("a0", 1), Operator.Eq, lit_i64(22)))))
    assert [r[0] for r in out] == h["x_having_max_y_22"]


def test_slt_count_and_bigint(engine, golden):
    c = golden["slt"]["count"]
    t = lambda v=c["v"]: _t(["v"], [(x,) for x in v])
    cnt = lambda inp: rows_of(engine.execute(q.NoGroupingAggregate(None, inp, [q.CountAggregateExpr(lit_i64(1))])))[0][0]
    assert cnt(t()) == c["count_all"]
    gt5 = lambda inp: _filter(inp, q.BinaryExpr(col("v", 0), Operator.Gt, lit_i64(5)))
    assert cnt(gt5(t())) == c["count_v_gt_5"]
    assert cnt(gt5(t([x for x in c["v"] if x != 7]))) == c["count_v_gt_5_after_delete_7"]
    assert cnt(_filter(t(), q.BinaryExpr(lit_i64(0), Operator.Eq, lit_i64(1)))) == c["count_where_false"]
    b = golden["slt"]["bigint"]
    t = _t(["v2"], [(x,) for x in b["v2"]])
    v2 = col("v2", 0)
    f = lambda op, lit: _filter(t, q.BinaryExpr(v2, op, lit_i64(lit)))
    assert cnt(f(Operator.Gt, 2)) == b["count_v2_gt_2"]
    assert rows_of(engine.execute(q.NoGroupingAggregate(None, f(Operator.Gt, 2), [q.MinAggregateExpr(v2, I64)]))) == [(b["min_where_gt_2"],)]
    assert rows_of(engine.execute(q.NoGroupingAggregate(None, t, [q.MaxAggregateExpr(v2, I64)]))) == [(b["max"],)]
    assert rows_of(engine.execute(q.NoGroupingAggregate(None, f(Operator.Lt, 10), [_sum(v2)]))) == [(b["sum_where_lt_10"],)]


def test_slt_join(engine, golden):
    j = golden["slt"]["join_xy"]
    x = _t(["a", "b"], j["x"])
    y = _t(["c", "d"], j["y"])
    plan = q.HashJoinExec.try_new(x, y, JoinType.Inner, [(col("a", 0), col("c", 0))], None)
    assert rows_of(engine.execute(plan)) == [tuple(r) for r in j["inner_a_eq_c"]]
    j = golden["slt"]["join_ab"]
    for jt, key in ((JoinType.Left, "left"), (JoinType.Right, "right"), (JoinType.Full, "full")):
        a = _t(["v1", "v2"], j["a"])
        b = _t(["v3", "v4"], j["b"])
        plan = q.HashJoinExec.try_new(a, b, jt, [(col("v1", 0), col("v3", 0))], None)
        assert rows_of(engine.execute(plan)) == [tuple(r) for r in j[key]], key
    j = golden["slt"]["join_two_keys"]
    a = _t(["v1", "v2"], j["a"])
    b = _t(["v3", "v4", "v5"], j["b"])
    on = [(col("v1", 0), col("v3", 0)), (col("v2", 1), col("v4", 1))]
    assert rows_of(engine.execute(q.HashJoinExec.try_new(a, b, JoinType.Inner, on, None))) == [tuple(r) for r in j["inner_v1_v3_and_v2_v4"]]
    fschema = pa.schema([pa.field("v1", I64), pa.field("v5", I64)])
    jf = q.JoinFilter(q.BinaryExpr(col("v1", 0), Operator.Lt, col("v5", 1)), [(0, q.JoinSide.Left), (2, q.JoinSide.Right)], fschema)
    assert rows_of(engine.execute(q.HashJoinExec.try_new(a, b, JoinType.Inner, on, jf))) == [tuple(r) for r in j["plus_residual_v1_lt_v5"]]


# ---------------------------------------------------------------- Q1 SF0.01 algebraic checks (q1.slt:24-27)
def test_q1_avg_is_truncating_division(engine, golden):
    """avg_qty == floor(sum_qty * 10^4 / count) * 10^-6 for all four groups pins DecimalAvgAccumulator (avg.rs:91-116)."""
    D = decimal.Decimal
    t = pa.decimal128(15, 2)
    for row in golden["tpch_q1_sf001"]["rows"]:
        sum_qty, avg_qty, count = D(row[2]), D(row[6]), row[9]
        # build a group whose SUM and COUNT equal the published ones: (count-1) rows of 0 plus one row holding the sum
        vals = pa.array([sum_qty] + [D("0.00")] * 3, type=t)
        # scale the check down: oracle's AVG of [sum, 0, 0, 0] with count rows is what we need -> use explicit accumulate
        batch = pa.RecordBatch.from_arrays([pa.array([1] * count, type=pa.int32()),
                                            pa.array([sum_qty] + [D("0.00")] * (count - 1), type=t)], names=["k", "q"])
        scan = table_scan(batch.schema, [batch])
        plan = q.HashAggregate(None, scan, [col("k", 0)], [q.AvgAggregateExpr(col("q", 1), t, q.avg_return_type(t))])
        out = engine.execute(plan)[0]
        assert out.column(1).type == pa.decimal128(19, 6)
        assert out.column(1)[0].as_py() == avg_qty, row[:2]


# ---------------------------------------------------------------- filter.slt / select.slt / basic_test.slt / type.slt (round 4)
_OPS = {"gt": Operator.Gt, "lt": Operator.Lt, "ge": Operator.GtEq, "le": Operator.LtEq, "eq": Operator.Eq, "and": Operator.And, "or": Operator.Or}


def _tree(node, names):
    """[operator, left, right] -> BinaryExpr in the SQL text's operand order (`3 > v1` keeps the literal on the left)"""
    if isinstance(node, str):
        return col(node, names.index(node))
    if isinstance(node, int):
        return lit_i64(node)
    return q.BinaryExpr(_tree(node[1], names), _OPS[node[0]], _tree(node[2], names))


def test_slt_filter(engine, golden):
    """tests/sql/filter.slt: nested AND / OR over comparisons with the literal on either side; the second table arrives in two
    inserts = two batches (Filter keeps one output batch per input batch, filter.rs:34)"""
    g = golden["slt"]["filter"]
    names = ["v1", "v2"]
    schema = pa.schema([pa.field(n, I64, False) for n in names])
    one_batch = lambda rows: pa.RecordBatch.from_arrays([pa.array([r[k] for r in rows], I64) for k in range(2)], schema=schema)   # noqa: E731
    for rows_key, queries, batches in (("t1", g["t1_queries"], [g["t1"]]), ("t2", g["t2_queries"], g["t2_batches"])):
        for case in queries:
            scan = table_scan(schema, [one_batch(b) for b in batches])
            k = names.index(case["select"])
            out = engine.execute(_filter(scan, _tree(case["where"], names)))
            assert len(out) == len(batches), case
            got = [r[k] for r in rows_of(out)]
            assert (sorted(got) if case["rowsort"] else got) == case["expect"], case


def test_slt_select_and_basic(engine, golden):
    """tests/sql/select.slt (projection arithmetic, aggregates with and without GROUP BY, a comparison of two aggregates projected
    over the aggregate) and basic_test.slt's MultiRowsMultiColumn"""
    g = golden["slt"]["select"]
    t = lambda: _t(["v1", "v2", "v3"], g["rows"])   # noqa: E731
    v1, v2, v3 = col("v1", 0), col("v2", 1), col("v3", 2)
    out = rows_of(engine.execute(q.Projection(None, t(), [q.BinaryExpr(v1, Operator.Add, v2)])))
    assert [r[0] for r in out] == g["v1_plus_v2"]
    assert rows_of(engine.execute(_agg(t(), [], [_sum(v1), _sum(v2)]))) == [tuple(g["sum_v1__sum_v2"])]
    # `group by v2, v2`: the planner keeps one key per distinct expression
    agg = q.HashAggregate(None, t(), [v2], [_sum(v1), q.CountAggregateExpr(v3), q.MinAggregateExpr(v3, I64), q.MaxAggregateExpr(v1, I64)])
    assert sorted(r[1:] for r in rows_of(engine.execute(agg))) == sorted(tuple(r) for r in g["group_by_v2__sum_v1__count_v3__min_v3__max_v1"])
    agg = q.HashAggregate(None, t(), [v2], [q.CountAggregateExpr(v3), q.MinAggregateExpr(v3, I64)])
    cmp = q.Projection(None, agg, [q.BinaryExpr(col("c1", 1), Operator.Eq, col("c2", 2))])
    assert sorted(r[0] for r in rows_of(engine.execute(cmp))) == g["group_by_v2__count_v3_eq_min_v3"]
    agg = q.HashAggregate(None, t(), [v2], [q.CountAggregateExpr(v3)])
    assert sorted((r[1], r[0]) for r in rows_of(engine.execute(agg))) == sorted(tuple(r) for r in g["group_by_v2__count_v3__v2"])
    assert sorted(r[0] for r in rows_of(engine.execute(q.HashAggregate(None, t(), [v2], [])))) == g["group_by_v2__v2"]
    out = rows_of(engine.execute(q.Projection(None, _filter(t(), q.BinaryExpr(v2, Operator.Gt, lit_i64(3))), [v1, v3])))
    assert sorted(out) == sorted(tuple(r) for r in g["v1_v3_where_v2_gt_3"])


def test_slt_type(engine, golden):
    """tests/sql/type.slt: Int16 arithmetic, a Date32 comparison against a literal date, ORDER BY a Boolean column, SimpleCaseExpr"""
    import datetime
    g = golden["slt"]["type"]
    a = col("a", 0)
    t = table_scan(pa.schema([pa.field("a", pa.int16(), False)]), [pa.RecordBatch.from_arrays([pa.array(g["smallint_a"], pa.int16())], names=["a"])])
    out = engine.execute(q.Projection(None, t, [q.BinaryExpr(a, op, a) for op in (Operator.Add, Operator.Sub, Operator.Mul, Operator.Div)]))
    assert rows_of(out) == [tuple(g["smallint_a_plus_minus_times_div_a"])] and all(f.type == pa.int16() for f in out[0].schema)
    days = [datetime.date.fromisoformat(s) for s in g["date_rows"]]
    t = table_scan(pa.schema([pa.field("v1", pa.date32(), False)]), [pa.RecordBatch.from_arrays([pa.array(days, pa.date32())], names=["v1"])])
    bound = q.CastExpr(q.Literal(S.Utf8(g["date_lt"])), pa.date32())
    out = rows_of(engine.execute(_filter(t, q.BinaryExpr(col("v1", 0), Operator.Lt, bound))))
    assert [r[0] for r in out] == [datetime.date.fromisoformat(s) for s in g["date_kept"]]
    # each INSERT is a batch of its own
    t = table_scan(pa.schema([pa.field("a", pa.bool_(), True)]), [pa.RecordBatch.from_arrays([pa.array([v], pa.bool_())], names=["a"]) for v in g["bool_rows"]])
    out = rows_of(engine.execute(q.Sort([q.PhysicalSortExpr(col("a", 0), q.SortOptions(False, True))], t)))
    assert [r[0] for r in out] == g["bool_order_by_asc"]
    t = table_scan(pa.schema([pa.field("x", I64, False)]), [pa.RecordBatch.from_arrays([pa.array(g["case_x"], I64)], names=["x"])])
    case = q.CaseExpr([(q.BinaryExpr(col("x", 0), Operator.Eq, lit_i64(1)), q.Literal(S.Utf8("a")))], q.Literal(S.Utf8("b")))
    assert sorted(r[0] for r in rows_of(engine.execute(q.Projection(None, t, [case])))) == g["case_x_when_1_then_a_else_b"]
