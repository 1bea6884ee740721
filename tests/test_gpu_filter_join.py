"""GPU parity: Filter / Scan(filter) and HashJoinExec (all join types, residual filter, NULL keys, duplicate keys,
multi-batch probe side) vs the reference's goldens and the CPU oracle — ordered, batch structure included."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator
from qurious_amd import ScalarValue as S

from .helpers import build_table_scan_i32, col, lit_i64, rows_of, table_scan

pytestmark = pytest.mark.gpu
I64 = pa.int64()


def _batches_equal(got, want):
    assert [b.num_rows for b in got] == [b.num_rows for b in want]
    assert rows_of(got) == rows_of(want)
    for g, w in zip(got, want):
        assert [f.type for f in g.schema] == [f.type for f in w.schema]


def _t(names, rows, types=None):
    types = types or [I64] * len(names)
    return table_scan(pa.schema([pa.field(n, t, True) for n, t in zip(names, types)]), [tuple(r) for r in rows])


# ---------------------------------------------------------------- Filter
def test_filter_slt_goldens(ctx, golden):
    g = golden["slt"]["where_t1"]
    t = _t(["v1", "v2"], g["rows"])
    v1, v2 = col("v1", 0), col("v2", 1)
    assert rows_of(q.Filter(t, q.BinaryExpr(v1, Operator.Gt, v2)).execute()) == [tuple(r) for r in g["v1_gt_v2"]]
    assert rows_of(q.Filter(t, q.BinaryExpr(v2, Operator.Gt, lit_i64(2))).execute()) == [tuple(r) for r in g["v2_gt_2"]]
    pred = q.BinaryExpr(q.BinaryExpr(v1, Operator.Eq, lit_i64(1)), Operator.Or, q.BinaryExpr(v2, Operator.Eq, lit_i64(2)))
    assert rows_of(q.Filter(t, pred).execute()) == [tuple(r) for r in g["v1_eq_1_or_v2_eq_2"]]
    g3 = golden["slt"]["where_t3"]
    t3 = _t(["v1", "v2"], g3["rows"])
    assert [r[1] for r in rows_of(q.Filter(t3, q.IsNull(v1)).execute())] == g3["v2_where_v1_is_null"]
    assert [r[1] for r in rows_of(q.Filter(t3, q.IsNotNull(v1)).execute())] == g3["v2_where_v1_is_not_null"]
    fn = golden["slt"]["filter_null"]
    assert rows_of(q.Filter(_t(["v1", "v2"], fn["rows"]), q.BinaryExpr(v1, Operator.Gt, lit_i64(1))).execute()) == [tuple(r) for r in fn["v1_gt_1"]]


def test_filter_all_layouts_multi_batch(ctx, oracle):
    """every Arrow layout on the path, NULLs in every column, ragged batches incl. empty ones; one output batch per input batch"""
    rng = np.random.default_rng(11)
    n = 5000
    D = decimal.Decimal
    dec = pa.decimal128(15, 2)
    def nul(a, t, p=0.1):
        return pa.array(a, type=t, mask=rng.random(len(a)) < p)
    words = ["", "A", "BUILDING", "x" * 70, "FURNITURE", "ab", "z" * 200]
    cols = [
        nul(rng.integers(-50, 50, n), pa.int32()),
        nul(rng.integers(-2**60, 2**60, n), I64),
        nul([D(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)], dec),
        nul([words[k] for k in rng.integers(0, len(words), n)], pa.string()),
        nul(rng.random(n) < 0.5, pa.bool_()),
        nul(rng.normal(size=n), pa.float64()),
        nul(rng.integers(8000, 11000, n), pa.int32()).cast(pa.date32()),
        nul(rng.integers(0, 255, n).astype(np.uint8), pa.uint8()),
    ]
    names = ["i32", "i64", "dec", "s", "b", "f", "d", "u8"]
    schema = pa.schema([pa.field(nm, c.type, True) for nm, c in zip(names, cols)])
    full = pa.RecordBatch.from_arrays(cols, schema=schema)
    cuts = [0, 0, 63, 64, 1000, 1000, 1001, 3333, 5000]
    batches = [full.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    scan = table_scan(schema, batches)
    preds = [
        q.BinaryExpr(col("i32", 0), Operator.Gt, q.Literal(S.Int32(0))),
        q.BinaryExpr(q.BinaryExpr(col("dec", 2), Operator.Lt, q.CastExpr(q.Literal(S.Int64(0)), dec)), Operator.Or, col("b", 4)),
        q.BinaryExpr(col("s", 3), Operator.Eq, q.Literal(S.Utf8("BUILDING"))),
        q.BinaryExpr(col("s", 3), Operator.Gt, q.Literal(S.Utf8("a"))),
        q.BinaryExpr(col("d", 6), Operator.LtEq, q.CastExpr(q.Literal(S.Utf8("1996-01-01")), pa.date32())),
        q.BinaryExpr(q.BinaryExpr(col("f", 5), Operator.Mul, q.Literal(S.Float64(2.0))), Operator.GtEq, q.Literal(S.Float64(0.5))),
        q.BinaryExpr(q.IsNull(col("u8", 7)), Operator.And, q.BinaryExpr(col("i64", 1), Operator.NotEq, lit_i64(7))),
        q.BinaryExpr(lit_i64(0), Operator.Eq, lit_i64(1)),
    ]
    for p in preds:
        plan = q.Filter(scan, p)
        _batches_equal(plan.execute(), oracle.execute(plan))
    # the pushed-down form (MemoryTable::scan, memory.rs:69-98)
    plan = q.Scan(schema, scan.datasource, None, preds[0])
    _batches_equal(plan.execute(), oracle.execute(plan))
    assert scan.datasource.scan(None, preds[2])[0].schema.names == names
    # projection + filter: the projection is applied first, the predicate indexes the projected batch (memory.rs:79-93)
    plan = q.Scan(pa.schema([schema.field("d"), schema.field("i32")]), scan.datasource, ["d", "i32"],
                  q.BinaryExpr(col("i32", 1), Operator.Lt, q.Literal(S.Int32(-10))))
    _batches_equal(plan.execute(), oracle.execute(plan))
    assert plan.execute()[0].schema.names == ["d", "i32"]



# ---------------------------------------------------------------- HashJoinExec
def _join_case(case):
    left = build_table_scan_i32(case["left"])
    right = build_table_scan_i32(case["right"])
    lnames, rnames = list(case["left"]), list(case["right"])
    on = [(col(lnames[l], l), col(rnames[r], r)) for l, r in case["on"]]
    return q.HashJoinExec.try_new(left, right, JoinType[case["join_type"]], on, None)


def test_hash_join_exec_reference_goldens(ctx, golden, join_layout):
    """hash_join.rs:396-698, 889-914 — output rows in the exact order the reference asserts"""
    for case in golden["hash_join_exec"]["cases"]:
        got = rows_of(_join_case(case).execute())
        assert got == [tuple(r) for r in case["expected"]], case["name"]


def test_slt_join_goldens(ctx, golden, join_layout):
    j = golden["slt"]["join_xy"]
    plan = q.HashJoinExec.try_new(_t(["a", "b"], j["x"]), _t(["c", "d"], j["y"]), JoinType.Inner, [(col("a", 0), col("c", 0))], None)
    assert rows_of(plan.execute()) == [tuple(r) for r in j["inner_a_eq_c"]]
    j = golden["slt"]["join_ab"]
    for jt, key in ((JoinType.Left, "left"), (JoinType.Right, "right"), (JoinType.Full, "full")):
        plan = q.HashJoinExec.try_new(_t(["v1", "v2"], j["a"]), _t(["v3", "v4"], j["b"]), jt, [(col("v1", 0), col("v3", 0))], None)
        assert rows_of(plan.execute()) == [tuple(r) for r in j[key]], key
    j = golden["slt"]["join_two_keys"]
    a, b = _t(["v1", "v2"], j["a"]), _t(["v3", "v4", "v5"], j["b"])
    on = [(col("v1", 0), col("v3", 0)), (col("v2", 1), col("v4", 1))]
    assert rows_of(q.HashJoinExec.try_new(a, b, JoinType.Inner, on, None).execute()) == [tuple(r) for r in j["inner_v1_v3_and_v2_v4"]]
    fschema = pa.schema([pa.field("v1", I64), pa.field("v5", I64)])
    jf = q.JoinFilter(q.BinaryExpr(col("v1", 0), Operator.Lt, col("v5", 1)), [(0, q.JoinSide.Left), (2, q.JoinSide.Right)], fschema)
    assert rows_of(q.HashJoinExec.try_new(a, b, JoinType.Inner, on, jf).execute()) == [tuple(r) for r in j["plus_residual_v1_lt_v5"]]


def _random_sides(rng, nl, nr, nkeys, null_p=0.08):
    def side(n, prefix):
        k1 = pa.array(rng.integers(0, nkeys, n), type=I64, mask=rng.random(n) < null_p)
        k2 = pa.array([("k%d" % v) for v in rng.integers(0, 3, n)], type=pa.string(), mask=rng.random(n) < null_p)
        pay = pa.array(rng.integers(0, 10**6, n), type=pa.int32(), mask=rng.random(n) < null_p)
        s = pa.array([("payload-%d" % v) * (1 + v % 3) for v in rng.integers(0, 50, n)], type=pa.string(), mask=rng.random(n) < null_p)
        names = [prefix + x for x in ("k1", "k2", "pay", "s")]
        schema = pa.schema([pa.field(nm, c.type, True) for nm, c in zip(names, (k1, k2, pay, s))])
        return schema, pa.RecordBatch.from_arrays([k1, k2, pay, s], schema=schema)
    return side(nl, "l_"), side(nr, "r_")


@pytest.mark.parametrize("jt", list(JoinType))
def test_hash_join_all_types_random_vs_oracle(ctx, oracle, jt, join_layout):
    """duplicate keys on both sides, NULL keys (never match), two-column keys (Int64 + Utf8), probe side in ragged batches"""
    rng = np.random.default_rng(100 + int(jt))
    (ls, lb), (rs, rb) = _random_sides(rng, 700, 1500, 60)
    left = table_scan(ls, [lb.slice(0, 300), lb.slice(300, 400)])
    cuts = [0, 64, 64, 700, 701, 1500]
    right = table_scan(rs, [rb.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    on = [(col("l_k1", 0), col("r_k1", 0)), (col("l_k2", 1), col("r_k2", 1))]
    plan = q.HashJoinExec.try_new(left, right, jt, on, None)
    _batches_equal(plan.execute(), oracle.execute(plan))
    # with a residual filter l_pay < r_pay
    fschema = pa.schema([pa.field("l_pay", pa.int32()), pa.field("r_pay", pa.int32())])
    jf = q.JoinFilter(q.BinaryExpr(col("l_pay", 0), Operator.Lt, col("r_pay", 1)), [(2, q.JoinSide.Left), (2, q.JoinSide.Right)], fschema)
    plan = q.HashJoinExec.try_new(left, right, jt, [on[0]], jf)
    _batches_equal(plan.execute(), oracle.execute(plan))


@pytest.mark.parametrize("force_csr", [False, True])
@pytest.mark.parametrize("unique_build", [True, False])
def test_hash_join_probe_many_tiles(ctx, oracle, monkeypatch, force_csr, unique_build, join_layout):
    """2.5 M probe rows (~10^4 probe tiles) with a fused scan filter, over unique build keys (slot -> row, also forced
    through the CSR path) and duplicated ones (CSR from the stable sort): the reference's pair order in every case."""
    if force_csr:
        monkeypatch.setenv("QHIP_JOIN_FORCE_CSR", "1")
    rng = np.random.default_rng(77)
    nb, npr = 60_000, 2_500_000
    bkeys = rng.permutation(nb * 4)[:nb] if unique_build else rng.integers(0, nb // 3, nb)
    ls = pa.schema([pa.field("b_key", I64), pa.field("b_pay", pa.int32())])
    rs = pa.schema([pa.field("p_key", I64), pa.field("p_pay", pa.int32()), pa.field("p_date", pa.date32())])
    lb = pa.RecordBatch.from_arrays([pa.array(bkeys, type=I64), pa.array(np.arange(nb), type=pa.int32())], schema=ls)
    pk = pa.array(rng.integers(0, nb * 4, npr), type=I64, mask=rng.random(npr) < 0.01)
    rb = pa.RecordBatch.from_arrays([pk, pa.array(np.arange(npr), type=pa.int32()),
                                     pa.array(rng.integers(9000, 9400, npr), type=pa.int32()).cast(pa.date32())], schema=rs)
    left = table_scan(ls, [lb])
    cuts = list(range(0, npr, 400_000)) + [npr]
    batches = [rb.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    pred = q.BinaryExpr(col("p_date", 2), Operator.Gt, q.Literal(S.Date32(9100)))
    on = [(col("b_key", 0), col("p_key", 0))]
    plan = q.HashJoinExec.try_new(left, table_scan(rs, batches, pred), JoinType.Inner, on, None)
    _batches_equal(plan.execute(), oracle.execute(plan))
    for jt in (JoinType.Left, JoinType.Right, JoinType.LeftSemi, JoinType.LeftAnti, JoinType.Full):
        plan = q.HashJoinExec.try_new(left, table_scan(rs, batches[:2]), jt, on, None)
        _batches_equal(plan.execute(), oracle.execute(plan))


def test_hash_join_many_pairs_per_probe_row(ctx, oracle, join_layout):
    """heavily duplicated keys on both sides: ~100 build rows per probe row"""
    rng = np.random.default_rng(78)
    (ls, lb), (rs, rb) = _random_sides(rng, 4000, 3000, 40, null_p=0.02)
    on = [(col("l_k1", 0), col("r_k1", 0))]
    for jt in (JoinType.Inner, JoinType.Left):
        plan = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb.slice(0, 1000), rb.slice(1000, 2000)]), jt, on, None)
        got = plan.execute()
        assert sum(b.num_rows for b in got) > 3000 * 10
        _batches_equal(got, oracle.execute(plan))


def test_hash_join_edge_cases(ctx, oracle, join_layout):
    (ls, lb), (rs, rb) = _random_sides(np.random.default_rng(5), 50, 80, 10)
    empty_l = table_scan(ls, [lb.slice(0, 0)])
    empty_r = table_scan(rs, [rb.slice(0, 0)])
    no_batches_r = q.Scan(rs, q.MemoryTable.try_new(rs, []))
    full_l, full_r = table_scan(ls, [lb]), table_scan(rs, [rb])
    on = [(col("l_k1", 0), col("r_k1", 0))]
    for jt in JoinType:
        for l, r in ((empty_l, full_r), (full_l, empty_r), (full_l, no_batches_r), (empty_l, empty_r)):
            plan = q.HashJoinExec.try_new(l, r, jt, on, None)
            _batches_equal(plan.execute(), oracle.execute(plan))
    with pytest.raises(q.QuriousError, match="should be non-empty"):
        q.HashJoinExec.try_new(full_l, full_r, JoinType.Inner, [], None)
    with pytest.raises(q.QuriousError, match="Invalid comparison operation"):
        q.HashJoinExec.try_new(full_l, full_r, JoinType.Inner, [(col("l_k1", 0), col("r_pay", 2))], None).execute()


def test_join_then_aggregate_stays_on_device(ctx, oracle, join_layout):
    """Q3-shaped mini pipeline: Scan(filter) |><| Scan(filter) -> HashAggregate(SUM(decimal expr)) — bit-exact"""
    rng = np.random.default_rng(9)
    D = decimal.Decimal
    dec = pa.decimal128(15, 2)
    no, nl = 3000, 12000
    o_schema = pa.schema([pa.field("o_orderkey", I64), pa.field("o_orderdate", pa.date32()), pa.field("o_shippriority", I64)])
    orders = pa.RecordBatch.from_arrays([pa.array(np.arange(1, no + 1), type=I64),
                                         pa.array(rng.integers(9000, 9400, no), type=pa.int32()).cast(pa.date32()),
                                         pa.array(np.zeros(no, dtype=np.int64))], schema=o_schema)
    l_schema = pa.schema([pa.field("l_orderkey", I64), pa.field("l_extendedprice", dec), pa.field("l_discount", dec), pa.field("l_shipdate", pa.date32())])
    li = pa.RecordBatch.from_arrays([pa.array(rng.integers(1, no + 1, nl), type=I64),
                                     pa.array([D(int(v)).scaleb(-2) for v in rng.integers(90100, 10**7, nl)], type=dec),
                                     pa.array([D(int(v)).scaleb(-2) for v in rng.integers(0, 11, nl)], type=dec),
                                     pa.array(rng.integers(9000, 9500, nl), type=pa.int32()).cast(pa.date32())], schema=l_schema)
    day = q.CastExpr(q.Literal(S.Utf8("1995-03-15")), pa.date32())
    o_scan = table_scan(o_schema, [orders], q.BinaryExpr(col("o_orderdate", 1), Operator.Lt, day))
    l_scan = table_scan(l_schema, [li.slice(k, 1024) for k in range(0, nl, 1024)], q.BinaryExpr(col("l_shipdate", 3), Operator.Gt, day))
    join = q.HashJoinExec.try_new(o_scan, l_scan, JoinType.Inner, [(col("o_orderkey", 0), col("l_orderkey", 0))], None)
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))
    revenue = q.BinaryExpr(col("l_extendedprice", 4), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("l_discount", 5)))
    agg = q.HashAggregate(None, join, [col("l_orderkey", 3), col("o_orderdate", 1), col("o_shippriority", 2)],
                          [q.SumAggregateExpr(revenue, pa.decimal128(38, 4))])
    got = sorted(rows_of(agg.execute()))
    want = sorted(rows_of(oracle.execute(agg)))
    assert got == want and len(got) > 100


def test_utf8_keys_longer_than_one_word(ctx, oracle, join_layout):
    """Utf8 group / join keys are packed into key words sized from the column's longest value"""
    rng = np.random.default_rng(77)
    segs = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY", "", "4-NOT SPECIFIED", "x" * 31, "y" * 16, "ab"]
    n = 20000
    schema = pa.schema([pa.field("seg", pa.string()), pa.field("v", I64)])
    batch = pa.RecordBatch.from_arrays([pa.array([segs[k] for k in rng.integers(0, len(segs), n)], type=pa.string(), mask=rng.random(n) < 0.05),
                                        pa.array(rng.integers(0, 1000, n), type=I64)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 9000), batch.slice(9000, 11000)])
    agg = q.HashAggregate(None, scan, [col("seg", 0)], [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(lit_i64(1))])
    assert sorted(rows_of(agg.execute()), key=repr) == sorted(rows_of(oracle.execute(agg)), key=repr)
    # join on the string key: right side has shorter strings only -> both sides still pack to the same width
    rs = pa.schema([pa.field("name", pa.string()), pa.field("w", I64)])
    rb = pa.RecordBatch.from_arrays([pa.array(["BUILDING", "ab", "MACHINERY", "zz", None, "BUILDING"]), pa.array([1, 2, 3, 4, 5, 6], type=I64)], schema=rs)
    for jt in (JoinType.Inner, JoinType.Left, JoinType.Full):
        plan = q.HashJoinExec.try_new(table_scan(rs, [rb]), scan, jt, [(col("name", 0), col("seg", 0))], None)
        _batches_equal(plan.execute(), oracle.execute(plan))
    long_schema = pa.schema([pa.field("s", pa.string())])
    long_scan = table_scan(long_schema, [pa.RecordBatch.from_arrays([pa.array(["z" * 56, "a"])], schema=long_schema)])
    with pytest.raises(q.UnsupportedError, match="longer than 55 bytes"):
        q.HashAggregate(None, long_scan, [col("s", 0)], [q.CountAggregateExpr(lit_i64(1))]).execute()


def test_utf8_keys_of_32_to_55_bytes(ctx, oracle, join_layout):
    """Round 4: a Utf8 key packs into up to 7 words (55 bytes + the length byte; TPC-H's names, addresses and l_comment fit) —
    values that differ only in their last byte, at every word boundary, NULLs and the empty string; GROUP BY alone and next to
    an Int64 key (1 + 7 = the 8 words a whole key may have), both sides of a join, every join layout."""
    rng = np.random.default_rng(91)
    base = "Customer#000000001 lives at 1 Long Street, Springfield!"          # 55 bytes
    assert len(base) == 55
    vals = [base, base[:-1] + "?", base[:54], base[:48], base[:47] + "x", base[:47] + "y", base[:40], base[:39], base[:32], base[:31], base[:33] + "é",
            "", "short", "héllo wörld " * 3]
    assert max(len(v.encode()) for v in vals) == 55
    n = 30000
    schema = pa.schema([pa.field("name", pa.string()), pa.field("k", I64), pa.field("v", I64)])
    batch = pa.RecordBatch.from_arrays([pa.array([vals[i] for i in rng.integers(0, len(vals), n)], type=pa.string(), mask=rng.random(n) < 0.04),
                                        pa.array(rng.integers(0, 3, n), type=I64), pa.array(rng.integers(-1000, 1000, n), type=I64)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 12000), batch.slice(12000)])
    aggs = [q.SumAggregateExpr(col("v", 2), I64), q.CountAggregateExpr(lit_i64(1)), q.MinAggregateExpr(col("v", 2), I64)]
    # (a whole key has at most 8 words: 7 of the name + the NULL-mask word here, 7 + the Int64 key over the NULL-free copy below)
    dense = pa.RecordBatch.from_arrays([pa.array([vals[i] for i in rng.integers(0, len(vals), n)], type=pa.string()), batch.column(1), batch.column(2)], schema=schema)
    for src, keys in ((scan, [col("name", 0)]), (table_scan(schema, [dense]), [col("k", 1), col("name", 0)])):
        agg = q.HashAggregate(None, src, keys, aggs)
        got = sorted(rows_of(agg.execute()), key=repr)
        assert got == sorted(rows_of(oracle.execute(agg)), key=repr) and len(got) >= len(vals)
    with pytest.raises(q.UnsupportedError, match="wider than 8 words"):
        q.HashAggregate(None, scan, [col("k", 1), col("name", 0)], aggs).execute()
    rs = pa.schema([pa.field("who", pa.string()), pa.field("w", I64)])
    rb = pa.RecordBatch.from_arrays([pa.array([base, base[:54], base[:47] + "y", "nobody " * 7, None, base, "short"]), pa.array(range(7), type=I64)], schema=rs)
    for jt in (JoinType.Inner, JoinType.Left, JoinType.Right, JoinType.Full, JoinType.LeftSemi, JoinType.LeftAnti):
        plan = q.HashJoinExec.try_new(table_scan(rs, [rb]), scan, jt, [(col("who", 0), col("name", 0))], None)
        _batches_equal(plan.execute(), oracle.execute(plan))
