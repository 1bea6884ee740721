"""GPU parity: fused filter + hash-aggregate HIP kernel vs the CPU oracle, through the C ABI (bit-exact)."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator, queries, synth
from qurious_amd import ScalarValue as S

from .helpers import col, lit_i64, rows_of, sorted_rows, table_scan

pytestmark = pytest.mark.gpu
I64 = pa.int64()


def _same(plan, oracle):
    got = sorted_rows(plan.execute())
    want = sorted_rows(oracle.execute(plan))
    assert got == want
    return got


def test_q1_mini_config0_matches_oracle(ctx, oracle):
    """configs[0]: 1M-row synthetic batch, l_shipdate < 1998-09-01 GROUP BY l_returnflag, SUM(l_quantity) — bit-exact."""
    table = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, synth.lineitem(1_000_000, 1_000_000))
    plan = queries.q1_mini(table)
    got = _same(plan, oracle)
    assert len(got) == 3
    out = plan.execute()
    assert out[0].schema.field(1).type == pa.decimal128(15, 2)
    st = ctx.last_stats()
    # (QHIP_AGG_PARTITION=2 in the environment forces the partitioned kernels on every grouped aggregate)
    assert st["main_kernel_name"] in ("qk_filter_agg", "qk_filter_agg_cons", "qk_agg_part_hist+scatter+reduce") and st["rows_in"] == 1_000_000


def test_q1_full_matches_oracle(ctx, oracle):
    """TPC-H Q1 aggregate list on 300k synthetic rows in 1024-row batches (the reference's COPY batch size)."""
    table = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, synth.lineitem(300_000, 1024))
    plan = queries.q1_full(table)
    got = _same(plan, oracle)
    assert len(got) == 4
    out = plan.execute()[0]
    assert [f.type for f in out.schema][2:] == [pa.decimal128(15, 2), pa.decimal128(15, 2), pa.decimal128(38, 4), pa.decimal128(38, 6),
                                                 pa.decimal128(19, 6), pa.decimal128(19, 6), pa.decimal128(19, 6), pa.int64()]


def _with_nulls(batch: pa.RecordBatch, rng, p=0.1) -> pa.RecordBatch:
    cols = []
    for k in range(batch.num_columns):
        mask = pa.array(rng.random(batch.num_rows) < p)
        cols.append(pa.compute.if_else(mask, pa.nulls(batch.num_rows, batch.column(k).type), batch.column(k)))
    schema = pa.schema([pa.field(f.name, f.type, True) for f in batch.schema])
    return pa.RecordBatch.from_arrays(cols, schema=schema)


def test_null_torture_1m_rows(ctx, oracle):
    """the 1 M-row null-torture batch — 10 % NULL in the date, key and value columns of synthetic lineitem —
    through the Q1 shapes: NULL predicates drop rows, NULL keys form their own groups, SUM / AVG / COUNT skip NULL values."""
    rng = np.random.default_rng(99)
    batches = [_with_nulls(b, rng) for b in synth.lineitem(1_000_000, 1 << 18)]
    schema = batches[0].schema
    table = q.MemoryTable.try_new(schema, batches)
    for plan in (queries.q1_mini(table), queries.q1_full(table)):
        got = _same(plan, oracle)
        assert any(r[0] is None for r in got)
    # the same data through a separate Filter node and an ungrouped aggregate
    pred = queries.q1_mini(table).input.filter
    scan = q.Scan(schema, table, None, None)
    agg = q.NoGroupingAggregate(None, q.Filter(scan, pred), [q.SumAggregateExpr(col("l_quantity", 3), pa.decimal128(15, 2)),
                                                             q.CountAggregateExpr(col("l_discount", 5)), q.MinAggregateExpr(col("l_shipdate", 0), pa.date32())])
    _same(agg, oracle)


def test_min_max_of_all_null_groups_yield_the_types_own_seed(ctx, oracle):
    """PrimitiveAccumulator is seeded with NATIVE::MAX / MIN (aggregate/mod.rs:60-84): a group whose values are all NULL
    reports the seed of the column's own type — i32::MAX for MIN(Int32), not a truncated i64::MAX (found by tools/fuzz_plans.py)"""
    k = pa.array([1, 1, 2, 2, 3], type=I64)
    cols = {"k": k, "i32": pa.array([None, None, 5, -7, None], type=pa.int32()), "i16": pa.array([None, None, 5, -7, None], type=pa.int16()),
            "d32": pa.array([None, None, 9000, 9001, None], type=pa.int32()).cast(pa.date32()), "i64": pa.array([None, None, 5, -7, None], type=I64)}
    schema = pa.schema([pa.field(n, a.type, True) for n, a in cols.items()])
    scan = table_scan(schema, [pa.RecordBatch.from_arrays(list(cols.values()), schema=schema)])
    aggs = []
    for idx, (name, arr) in enumerate(list(cols.items())[1:], start=1):
        aggs += [q.MinAggregateExpr(col(name, idx), arr.type), q.MaxAggregateExpr(col(name, idx), arr.type)]
    for plan in (q.HashAggregate(None, scan, [col("k", 0)], aggs),):
        got = {r[0]: r[1:] for r in rows_of([b.cast(pa.schema([pa.field(f.name, pa.int32() if pa.types.is_date32(f.type) else f.type) for f in b.schema])) for b in plan.execute()])}
        want = {r[0]: r[1:] for r in rows_of([b.cast(pa.schema([pa.field(f.name, pa.int32() if pa.types.is_date32(f.type) else f.type) for f in b.schema])) for b in oracle.execute(plan)])}
        assert got == want
        assert got[1] == (2**31 - 1, -2**31, 2**15 - 1, -2**15, 2**31 - 1, -2**31, 2**63 - 1, -2**63)
        assert got[2] == (-7, 5, -7, 5, 9000, 9001, -7, 5)


def test_group_by_int64_with_nulls_and_expr_keys(ctx, oracle):
    rng = np.random.default_rng(3)
    n = 20000
    k = rng.integers(0, 37, n)
    v = rng.integers(-10**12, 10**12, n)
    kmask = rng.random(n) < 0.1
    vmask = rng.random(n) < 0.1
    schema = pa.schema([pa.field("k", I64), pa.field("v", I64)])
    batch = pa.RecordBatch.from_arrays([pa.array(k, type=I64, mask=kmask), pa.array(v, type=I64, mask=vmask)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 7000), batch.slice(7000, 13000)])
    key = q.BinaryExpr(col("k", 0), Operator.Add, lit_i64(1))
    aggs = [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(col("v", 1)), q.CountAggregateExpr(lit_i64(1)),
            q.MinAggregateExpr(col("v", 1), I64), q.MaxAggregateExpr(col("v", 1), I64)]
    _same(q.HashAggregate(None, scan, [key], aggs), oracle)
    pred = q.BinaryExpr(col("v", 1), Operator.Gt, lit_i64(0))
    _same(q.HashAggregate(None, q.Filter(scan, pred), [col("k", 0)], aggs), oracle)


def test_high_cardinality_group_by(ctx, oracle):
    """more groups than the LDS-staged table holds: keys spill to the HBM table, table growth retry path"""
    rng = np.random.default_rng(5)
    n = 400_000
    k = rng.integers(0, 150_000, n)
    v = rng.integers(0, 1000, n)
    schema = pa.schema([pa.field("k", I64), pa.field("v", I64)])
    scan = table_scan(schema, [pa.RecordBatch.from_arrays([pa.array(k, type=I64), pa.array(v, type=I64)], schema=schema)])
    plan = q.HashAggregate(None, scan, [col("k", 0)], [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(lit_i64(1))])
    got = _same(plan, oracle)
    assert len(got) == len(np.unique(k))


def test_decimal_i128_wraparound_sum_is_exact(ctx, oracle):
    """values near +-2^126 so that low-half carries and 128-bit wrap-around both happen (SUM is add_wrapping, sum.rs:75-77)"""
    t = pa.decimal128(38, 0)
    big = (1 << 126) - 12345
    vals = [big, big, big, -big, 7, (1 << 64) - 1, (1 << 64) - 1, -(1 << 100)] * 500
    D = decimal.Decimal
    ctx38 = decimal.Context(prec=60)
    raw = np.array([[v & 0xFFFFFFFFFFFFFFFF, (v >> 64) & 0xFFFFFFFFFFFFFFFF] for v in vals], dtype=np.uint64)
    arr = pa.Array.from_buffers(t, len(vals), [None, pa.py_buffer(raw)])
    keys = pa.array([i % 3 for i in range(len(vals))], type=pa.int32())
    schema = pa.schema([pa.field("k", pa.int32()), pa.field("v", t)])
    scan = table_scan(schema, [pa.RecordBatch.from_arrays([keys, arr], schema=schema)])
    plan = q.HashAggregate(None, scan, [col("k", 0)], [q.SumAggregateExpr(col("v", 1), t)])
    got = plan.execute()[0]
    want = oracle.execute(plan)[0]
    to_raw = lambda b: sorted((b.column(0)[i].as_py(), bytes(b.column(1).buffers()[1])[16 * i:16 * i + 16]) for i in range(b.num_rows))
    assert to_raw(got) == to_raw(want)


def test_no_grouping_and_empty_inputs(ctx, oracle):
    schema = pa.schema([pa.field("v1", I64), pa.field("v2", pa.float64())])
    rows = [(1, 2.5), (2, 3.2), (None, 4.7), (4, None)]
    scan = table_scan(schema, rows)
    aggs = [q.SumAggregateExpr(col("v1", 0), I64), q.SumAggregateExpr(col("v2", 1), pa.float64()), q.CountAggregateExpr(col("v1", 0)),
            q.MinAggregateExpr(col("v1", 0), I64), q.MaxAggregateExpr(col("v2", 1), pa.float64()),
            q.AvgAggregateExpr(col("v2", 1), pa.float64(), pa.float64())]
    got = rows_of(q.NoGroupingAggregate(None, scan, aggs).execute())
    want = rows_of(oracle.execute(q.NoGroupingAggregate(None, scan, aggs)))
    assert len(got) == 1 and got[0][0] == want[0][0] and got[0][2:4] == want[0][2:4]
    for a, b in ((got[0][1], want[0][1]), (got[0][4], want[0][4]), (got[0][5], want[0][5])):
        assert abs(a - b) <= 1e-6 * abs(b)
    # zero batches: COUNT -> 0, SUM -> NULL; grouped: no output batches (aggregation.slt:162-169,192-195)
    empty = q.Scan(schema, q.MemoryTable.try_new(schema, []))
    assert rows_of(q.NoGroupingAggregate(None, empty, [q.CountAggregateExpr(lit_i64(1)), q.SumAggregateExpr(col("v1", 0), I64)]).execute()) == [(0, None)]
    assert q.HashAggregate(None, empty, [col("v1", 0)], [q.CountAggregateExpr(lit_i64(1))]).execute() == []
    # a filter that keeps nothing: groups = 0 rows in one batch
    none = q.Filter(scan, q.BinaryExpr(col("v1", 0), Operator.Gt, lit_i64(100)))
    out = q.HashAggregate(None, none, [col("v1", 0)], [q.CountAggregateExpr(lit_i64(1))]).execute()
    assert len(out) == 1 and out[0].num_rows == 0


def test_errors_match_reference_behaviour(ctx, oracle):
    schema = pa.schema([pa.field("a", I64), pa.field("b", pa.int32()), pa.field("f", pa.float64())])
    scan = table_scan(schema, [(1, 1, 1.0), (2, 0, 2.0)])
    with pytest.raises(q.QuriousError, match="Invalid comparison operation"):
        q.HashAggregate(None, q.Filter(scan, q.BinaryExpr(col("a", 0), Operator.Eq, col("b", 1))), [col("a", 0)], []).execute()
    with pytest.raises(q.QuriousError, match="Unsupported data type in hasher"):
        q.HashAggregate(None, scan, [col("f", 2)], [q.CountAggregateExpr(lit_i64(1))]).execute()
    with pytest.raises(q.QuriousError, match="Divide by zero"):
        q.HashAggregate(None, scan, [col("a", 0)], [q.SumAggregateExpr(q.BinaryExpr(col("a", 0), Operator.Div, q.CastExpr(col("b", 1), I64)), I64)]).execute()
    with pytest.raises(q.QuriousError, match="Sum not supported"):
        q.HashAggregate(None, scan, [col("a", 0)], [q.SumAggregateExpr(col("b", 1), pa.int32())]).execute()


def test_device_side_output_assembly_matches_host_side(ctx, oracle, monkeypatch):
    """many groups are finished on the device (k_agg_finalize); force that path on small inputs and compare every output kind"""
    monkeypatch.setenv("QHIP_AGG_DEVICE_FINALIZE_MIN_GROUPS", "1")
    monkeypatch.setenv("QHIP_AGG_REPLICAS", "1")
    rng = np.random.default_rng(17)
    n = 30000
    D = decimal.Decimal
    dec = pa.decimal128(15, 2)
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("d", dec), pa.field("v", I64), pa.field("f", pa.float64()),
                        pa.field("dt", pa.date32()), pa.field("x", dec)])
    batch = pa.RecordBatch.from_arrays([
        pa.array(rng.integers(0, 500, n), type=I64, mask=rng.random(n) < 0.05),
        pa.array(["g%d" % v for v in rng.integers(0, 30, n)], type=pa.string(), mask=rng.random(n) < 0.05),
        pa.array([D(int(v)).scaleb(-2) for v in rng.integers(0, 4, n)], type=dec),
        pa.array(rng.integers(-10**9, 10**9, n), type=I64, mask=rng.random(n) < 0.1),
        pa.array(rng.normal(size=n), type=pa.float64(), mask=rng.random(n) < 0.1),
        pa.array(rng.integers(9000, 9100, n), type=pa.int32()).cast(pa.date32()),
        pa.array([D(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)], type=dec, mask=rng.random(n) < 0.1)], schema=schema)
    scan = table_scan(schema, [batch])
    aggs = [q.SumAggregateExpr(col("v", 3), I64), q.CountAggregateExpr(col("v", 3)), q.CountAggregateExpr(lit_i64(1)),
            q.MinAggregateExpr(col("v", 3), I64), q.MaxAggregateExpr(col("dt", 5), pa.date32()),
            q.SumAggregateExpr(col("x", 6), dec), q.AvgAggregateExpr(col("x", 6), dec, q.avg_return_type(dec)),
            q.MinAggregateExpr(col("x", 6), dec), q.MaxAggregateExpr(col("x", 6), dec)]
    plan = q.HashAggregate(None, scan, [col("k", 0), col("s", 1), col("d", 2)], aggs)
    got, want = plan.execute()[0], oracle.execute(plan)[0]
    assert [f.type for f in got.schema] == [f.type for f in want.schema]
    assert sorted_rows([got]) == sorted_rows([want]) and got.num_rows > 5000
    fl = q.HashAggregate(None, scan, [col("k", 0)], [q.SumAggregateExpr(col("f", 4), pa.float64()),
                                                     q.AvgAggregateExpr(col("f", 4), pa.float64(), pa.float64()),
                                                     q.MinAggregateExpr(col("f", 4), pa.float64()), q.MaxAggregateExpr(col("f", 4), pa.float64())])
    g2 = {r[0]: r[1:] for r in rows_of(fl.execute())}
    w2 = {r[0]: r[1:] for r in rows_of(oracle.execute(fl))}
    assert g2.keys() == w2.keys()
    for k in g2:
        for a, b in zip(g2[k], w2[k]):
            assert (a is None and b is None) or abs(a - b) <= 1e-6 * max(1.0, abs(b))


def test_partitioned_path_for_many_groups(ctx, oracle, monkeypatch):
    """the partitioned aggregate (rows split by key-hash bin, per-bin LDS aggregation, one HBM merge per group — what runs for
    many groups on a big input) forced on: every cell kind, NULL keys and values, two keys incl. Utf8, a fused filter,
    heavy keys next to a long tail, more groups than one LDS table per bin holds"""
    import decimal
    monkeypatch.setenv("QHIP_AGG_PARTITION", "2")
    rng = np.random.default_rng(41)
    n = 300_000
    heavy = rng.random(n) < 0.3
    k = np.where(heavy, rng.integers(0, 5, n), rng.integers(0, 120_000, n))
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("v", I64), pa.field("d", pa.decimal128(15, 2)), pa.field("f", pa.float64())])
    batch = pa.RecordBatch.from_arrays([
        pa.array(k, type=I64, mask=rng.random(n) < 0.02),
        pa.array(["g%d" % v for v in rng.integers(0, 7, n)], type=pa.string(), mask=rng.random(n) < 0.02),
        pa.array(rng.integers(-10**9, 10**9, n), type=I64, mask=rng.random(n) < 0.1),
        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**10, 10**10, n)], type=pa.decimal128(15, 2), mask=rng.random(n) < 0.1),
        pa.array(rng.integers(-1000, 1000, n).astype(np.float64), type=pa.float64(), mask=rng.random(n) < 0.1)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 100_000), batch.slice(100_000, 0), batch.slice(100_000)])
    D = pa.decimal128(15, 2)
    aggs = [q.SumAggregateExpr(col("v", 2), I64), q.CountAggregateExpr(col("v", 2)), q.CountAggregateExpr(lit_i64(1)),
            q.MinAggregateExpr(col("v", 2), I64), q.MaxAggregateExpr(col("d", 3), D), q.SumAggregateExpr(col("d", 3), D),
            q.AvgAggregateExpr(col("d", 3), D, pa.decimal128(19, 6)), q.SumAggregateExpr(col("f", 4), pa.float64()),
            q.MinAggregateExpr(col("f", 4), pa.float64())]
    got = _same(q.HashAggregate(None, scan, [col("k", 0)], aggs), oracle)
    assert len(got) > 90_000
    assert ctx.last_stats()["main_kernel_name"].startswith("qk_agg_part")
    _same(q.HashAggregate(None, scan, [col("k", 0), col("s", 1)], aggs[:4]), oracle)
    pred = q.BinaryExpr(col("v", 2), Operator.Gt, lit_i64(0))
    _same(q.HashAggregate(None, q.Scan(schema, scan.datasource, None, pred), [col("k", 0)], aggs[:3]), oracle)
    # the automatic choice: a plan that produced many groups from a big input partitions its next run by itself
    monkeypatch.delenv("QHIP_AGG_PARTITION")
    m = 5_000_000
    kk = rng.integers(0, 400_000, m)
    big = pa.schema([pa.field("k", I64), pa.field("v", I64)])
    bscan = table_scan(big, [pa.RecordBatch.from_arrays([pa.array(kk, type=I64), pa.array(rng.integers(0, 100, m), type=I64)], schema=big)])
    plan = q.HashAggregate(None, bscan, [col("k", 0)], [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(lit_i64(1))])
    first = sorted(rows_of(plan.execute()))
    assert not ctx.last_stats()["main_kernel_name"].startswith("qk_agg_part")
    second = sorted(rows_of(plan.execute()))
    assert ctx.last_stats()["main_kernel_name"].startswith("qk_agg_part")
    assert first == second and len(first) == len(np.unique(kk))
    sums = np.bincount(kk, weights=None, minlength=400_000)
    assert [r[2] for r in first] == [int(c) for c in sums[sums > 0]]


def test_wide_workgroups_for_mid_sized_many_group_inputs(ctx, monkeypatch):
    """A plan that produced many groups (> 4096) from a mid-sized input (1-4 M rows) runs its NEXT execution with one 1024-thread
    workgroup per CU on one LDS table of up to 128 KB (qk_filter_agg_wide, csrc/agg.cpp `wide`). Every cell kind — 128-bit and
    narrow SUMs, COUNT, 64-bit and 128-bit MIN / MAX (the per-cell lock), Float64 SUM — two key columns incl. Utf8, NULL keys and
    values, a fused filter, heavy keys next to a long tail; compared with the 256-thread shape (QHIP_AGG_WIDE=0), with the first
    execution (which does not know the group count yet) and with exact numpy results for the integer columns."""
    import decimal
    rng = np.random.default_rng(77)
    n = 1_300_000
    heavy = rng.random(n) < 0.25
    k = np.where(heavy, rng.integers(0, 6, n), rng.integers(0, 90_000, n))
    kn = rng.random(n) < 0.01
    v = rng.integers(-10**9, 10**9, n)
    vn = rng.random(n) < 0.05
    d_raw = rng.integers(-10**12, 10**12, n)
    dn = rng.random(n) < 0.05
    D = pa.decimal128(20, 2)
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("v", I64), pa.field("d", D), pa.field("f", pa.float64())])
    batch = pa.RecordBatch.from_arrays([
        pa.array(k, type=I64, mask=kn),
        pa.array(["s%d" % x for x in rng.integers(0, 3, n)], type=pa.string()),
        pa.array(v, type=I64, mask=vn),
        pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in d_raw], type=D, mask=dn),
        pa.array(rng.integers(-100, 100, n).astype(np.float64), type=pa.float64())], schema=schema)
    cuts = [0, 400_000, 400_000, 900_001, n]
    scan = q.Scan(schema, q.MemoryTable.try_new(schema, [batch.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]), None,
                  q.BinaryExpr(col("f", 4), Operator.GtEq, q.Literal(q.ScalarValue.Float64(-90.0))))
    aggs = [q.SumAggregateExpr(col("v", 2), I64), q.CountAggregateExpr(col("v", 2)), q.CountAggregateExpr(lit_i64(1)), q.MinAggregateExpr(col("v", 2), I64),
            q.MaxAggregateExpr(col("d", 3), D), q.MinAggregateExpr(col("d", 3), D), q.SumAggregateExpr(col("d", 3), D), q.SumAggregateExpr(col("f", 4), pa.float64())]
    plan = q.HashAggregate(None, scan, [col("k", 0), col("s", 1)], aggs)

    def run():
        out = sorted(rows_of(plan.execute()), key=lambda r: tuple((x is None, x) for x in r[:2]))
        return out, ctx.last_stats()

    first, st1 = run()
    assert st1["workgroups"] > 256 and st1["groups"] > 90_000          # the default shape: the group count is not known yet
    second, st2 = run()
    assert st2["workgroups"] <= 256 and st2["lds_table_slots"] >= 256    # one workgroup per CU, the big table (128 KB / slot bytes)
    monkeypatch.setenv("QHIP_AGG_WIDE", "0")
    third, st3 = run()
    assert st3["workgroups"] <= 256 and st3["lds_table_slots"] < st2["lds_table_slots"]
    close = lambda a, b: len(a) == len(b) and all(x[:9] == y[:9] and abs(x[9] - y[9]) <= 1e-6 * max(1.0, abs(y[9])) for x, y in zip(a, b))   # noqa: E731
    assert close(first, second) and close(third, second)
    # exact integers of one heavy and a few light groups
    keep = np.asarray(batch.column(4).to_pylist()) >= -90.0
    s_col = np.asarray(batch.column(1).to_pylist())
    for key in (0, 3, int(k[~heavy][0]), int(k[~heavy][5])):
        for sv in ("s0", "s2"):
            m = keep & ~kn & (k == key) & (s_col == sv)
            row = [r for r in second if r[0] == key and r[1] == sv]
            if not m.any():
                assert not row
                continue
            assert len(row) == 1
            mv = m & ~vn
            assert row[0][2] == (int(v[mv].sum()) if mv.any() else None) and row[0][3] == int(mv.sum()) and row[0][4] == int(m.sum())
            assert row[0][5] == (int(v[mv].min()) if mv.any() else None)
            md = m & ~dn
            assert row[0][6] == (decimal.Decimal(int(d_raw[md].max())).scaleb(-2) if md.any() else None)
            assert row[0][7] == (decimal.Decimal(int(d_raw[md].min())).scaleb(-2) if md.any() else None)
            assert row[0][8] == (decimal.Decimal(int(d_raw[md].sum())).scaleb(-2) if md.any() else None)


def test_prepartitioned_aggregate_of_mid_sized_many_group_inputs(ctx, oracle, monkeypatch):
    """Round 4 (csrc/agg.cpp AggParts): a mid-sized input with many groups is ordered by key hash into one part per workgroup
    first (the exchange's partition passes over the row numbers), every part is aggregated in its workgroup's LDS table alone
    and appended to the result — nothing merged into the HBM table; a part holding a heavy key is sliced and its slices merge
    like the unpartitioned kernel. Forced on small inputs here: every cell kind, NULL keys and values, Utf8 + Int64 keys, heavy
    keys next to a long tail (sliced parts), a single group, more groups than the parts' LDS tables hold; then the automatic
    choice on 2 M rows -> 200 k groups."""
    import decimal
    monkeypatch.setenv("QHIP_AGG_PARTS", "2")
    rng = np.random.default_rng(43)
    n = 260_000
    heavy = rng.random(n) < 0.4
    k = np.where(heavy, rng.integers(0, 3, n), rng.integers(0, 100_000, n))
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("v", I64), pa.field("d", pa.decimal128(15, 2)), pa.field("f", pa.float64())])
    batch = pa.RecordBatch.from_arrays([
        pa.array(k, type=I64, mask=rng.random(n) < 0.02),
        pa.array(["g%d" % v for v in rng.integers(0, 7, n)], type=pa.string(), mask=rng.random(n) < 0.02),
        pa.array(rng.integers(-10**9, 10**9, n), type=I64, mask=rng.random(n) < 0.1),
        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**10, 10**10, n)], type=pa.decimal128(15, 2), mask=rng.random(n) < 0.1),
        pa.array(rng.integers(-1000, 1000, n).astype(np.float64), type=pa.float64(), mask=rng.random(n) < 0.1)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 100_000), batch.slice(100_000, 0), batch.slice(100_000)])
    D = pa.decimal128(15, 2)
    aggs = [q.SumAggregateExpr(col("v", 2), I64), q.CountAggregateExpr(col("v", 2)), q.CountAggregateExpr(lit_i64(1)),
            q.MinAggregateExpr(col("v", 2), I64), q.MaxAggregateExpr(col("d", 3), D), q.SumAggregateExpr(col("d", 3), D),
            q.AvgAggregateExpr(col("d", 3), D, pa.decimal128(19, 6)), q.SumAggregateExpr(col("f", 4), pa.float64()),
            q.MinAggregateExpr(col("f", 4), pa.float64())]
    plan = q.HashAggregate(None, scan, [col("k", 0)], aggs)
    for _ in range(3):   # (the third execution knows the groups: as many parts as the LDS tables need, heavy parts sliced)
        got = _same(plan, oracle)
        assert "qk_filter_agg_parts" in ctx.last_stats()["main_kernel_name"]
    assert len(got) > 70_000
    _same(q.HashAggregate(None, scan, [col("k", 0), col("s", 1)], aggs[:4]), oracle)
    _same(q.HashAggregate(None, scan, [col("s", 1)], aggs), oracle)                      # 8 groups: every part but a few empty
    one = pa.RecordBatch.from_arrays([pa.array(np.zeros(5000, dtype=np.int64)), pa.array(np.arange(5000, dtype=np.int64))], names=["k", "v"])
    _same(q.HashAggregate(None, table_scan(one.schema, [one]), [col("k", 0)], [q.SumAggregateExpr(col("v", 1), I64)]), oracle)
    # the automatic choice: 2 M rows of plain columns that produced 200 k groups last time (what the context learnt about this
    # aggregate — identified by its expressions — in earlier tests is forgotten first)
    monkeypatch.delenv("QHIP_AGG_PARTS")
    ctx.forget_plans()
    m = 2_000_000
    kk = rng.integers(0, 200_000, m)
    vv = rng.integers(0, 100, m)
    big = pa.schema([pa.field("k", I64), pa.field("v", I64)])
    bscan = table_scan(big, [pa.RecordBatch.from_arrays([pa.array(kk, type=I64), pa.array(vv, type=I64)], schema=big)])
    bplan = q.HashAggregate(None, bscan, [col("k", 0)], [q.SumAggregateExpr(col("v", 1), I64), q.CountAggregateExpr(lit_i64(1))])
    first = sorted(rows_of(bplan.execute()))
    assert "parts" not in ctx.last_stats()["main_kernel_name"]
    second = sorted(rows_of(bplan.execute()))
    assert "qk_filter_agg_parts" in ctx.last_stats()["main_kernel_name"]
    assert first == second
    cnt = np.bincount(kk, minlength=200_000)
    sm = np.bincount(kk, weights=vv, minlength=200_000)
    assert [(r[0], r[1], r[2]) for r in first] == [(int(g), int(sm[g]), int(cnt[g])) for g in np.nonzero(cnt)[0]]


def test_remembered_group_count_from_a_smaller_table_does_not_truncate_the_result(ctx):
    """What the context learns about an aggregate is keyed by its expressions, not by its table: the same query over a much
    bigger table starts with the small table's group count. The dense buffer is then too small, the slots are compacted again
    with the exact size — and the output columns the speculative device-side assembly prepared for the OLD size must not be
    taken as the result (round 4: found by Q3 at SF 0.5 followed by Q3 at SF 10 in one process)."""
    rng = np.random.default_rng(7)
    schema = pa.schema([pa.field("k", I64), pa.field("v", I64)])

    def plan_over(n, groups):
        k, v = rng.integers(0, groups, n), rng.integers(0, 1000, n)
        batch = pa.RecordBatch.from_arrays([pa.array(k, type=I64), pa.array(v, type=I64)], schema=schema)
        want = np.bincount(k, weights=v.astype(np.float64), minlength=groups)
        return q.HashAggregate(None, table_scan(schema, [batch]), [col("k", 0)], [q.SumAggregateExpr(col("v", 1), I64)]), k, want

    ctx.forget_plans()
    small, _, _ = plan_over(200_000, 6_000)
    for _ in range(2):                      # learnt: ~6 000 groups, assembled on the device from the second execution on
        assert len(rows_of(small.execute())) == 6_000
    big, k, want = plan_over(3_000_000, 150_000)
    for _ in range(2):
        got = rows_of(big.execute())
        present = np.unique(k)
        assert len(got) == len(present)
        got_k = np.array([r[0] for r in got]); got_v = np.array([r[1] for r in got], dtype=np.float64)
        assert (np.sort(got_k) == present).all() and (got_v == want[got_k]).all()
    assert len(rows_of(small.execute())) == 6_000     # ... and back: a remembered count far above the truth


def test_consecutive_rows_form_of_the_fused_kernel(ctx, oracle, monkeypatch):
    """QHIP_AGG_CONS=2 (forced; automatic only for register-light plans: measured slower on Q1): a lane owns four consecutive rows of a tile and
    loads a column's four values at once; the last, partial tile is shifted back and the rows it shares with the tile before
    are masked out. Q1's shape (two one-byte string keys, narrow decimal copies, a date filter) over row counts around the
    tile size, evaluated 1 / 2 / 4 rows at a time, with and without the second register set."""
    import decimal
    rng = np.random.default_rng(3)
    schema = pa.schema([pa.field("f", pa.string()), pa.field("s", pa.string()), pa.field("d", pa.date32()), pa.field("q", pa.decimal128(15, 2)),
                        pa.field("p", pa.decimal128(15, 2)), pa.field("i", pa.int32())])

    def batch_of(n):
        return pa.RecordBatch.from_arrays([
            pa.array([("A", "N", "R")[v] for v in rng.integers(0, 3, n)]), pa.array([("F", "O")[v] for v in rng.integers(0, 2, n)]),
            pa.array(rng.integers(9000, 10600, n).astype(np.int32), type=pa.int32()).cast(pa.date32()),
            pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(100, 5001, n)], type=pa.decimal128(15, 2)),
            pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-90000, 10**7, n)], type=pa.decimal128(15, 2)),
            pa.array(rng.integers(-5, 5, n), type=pa.int32())], schema=schema)

    D = pa.decimal128(15, 2)
    pred = q.BinaryExpr(col("d", 2), Operator.LtEq, q.CastExpr(q.Literal(q.ScalarValue.Int32(10400)), pa.date32()))
    monkeypatch.setenv("QHIP_AGG_CONS", "2")
    monkeypatch.setenv("QHIP_STATS_MIN_ROWS", "1")          # narrow copies from the first read on
    for sb, pipe in (("1", "0"), ("4", "1")):
        monkeypatch.setenv("QHIP_AGG_CONS_SB", sb)
        monkeypatch.setenv("QHIP_AGG_CONS_PIPE", pipe)
        ctx.forget_plans()
        for n in (65_536, 65_537, 262_144 + 1023, 300_001):
            b = batch_of(n)
            scan = table_scan(schema, [b.slice(0, n // 3), b.slice(n // 3)], pred)
            plan = q.HashAggregate(None, scan, [col("f", 0), col("s", 1)],
                                   [q.SumAggregateExpr(col("q", 3), D), q.SumAggregateExpr(col("p", 4), D), q.CountAggregateExpr(lit_i64(1)),
                                    q.SumAggregateExpr(q.BinaryExpr(col("q", 3), Operator.Mul, col("p", 4)), pa.decimal128(31, 4))])
            for _ in range(2):
                got = _same(plan, oracle)
            assert ctx.last_stats()["main_kernel_name"] == "qk_filter_agg_cons" and len(got) == 6, (sb, pipe, n)
        # an Int32 key next to the flags (three key words), MIN / MAX cells
        plan = q.HashAggregate(None, table_scan(schema, [batch_of(100_000)]), [col("i", 5), col("f", 0)],
                               [q.MinAggregateExpr(col("p", 4), D), q.MaxAggregateExpr(col("q", 3), D), q.CountAggregateExpr(col("p", 4))])
        _same(plan, oracle)
        assert ctx.last_stats()["main_kernel_name"] == "qk_filter_agg_cons"
    ctx.forget_plans()


def _runs_table(rng, n, max_run, with_nulls=True):
    """keys in non-decreasing order with runs of 1..max_run rows; a second key column that depends on the first; values of every kind"""
    import decimal
    lens = rng.integers(1, max_run + 1, n)
    k = np.repeat(np.arange(len(lens)), lens)[:n] * 3 + 5
    schema = pa.schema([pa.field("k", I64), pa.field("d", pa.date32()), pa.field("v", I64), pa.field("p", pa.decimal128(15, 2)), pa.field("f", pa.float64()),
                        pa.field("s", pa.string())])
    vmask = (rng.random(n) < 0.1) if with_nulls else None
    batch = pa.RecordBatch.from_arrays([
        pa.array(k, type=I64), pa.array((9000 + k % 700).astype(np.int32), type=pa.int32()).cast(pa.date32()),
        pa.array(rng.integers(-10**6, 10**6, n), type=I64, mask=vmask),
        pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**9, 10**9, n)], type=pa.decimal128(15, 2), mask=vmask),
        pa.array(rng.integers(-500, 500, n).astype(np.float64), type=pa.float64()),
        pa.array(["k%07d" % x for x in k], type=pa.string())], schema=schema)
    return schema, batch


def _runs_aggs():
    D = pa.decimal128(15, 2)
    return [q.SumAggregateExpr(col("v", 2), I64), q.CountAggregateExpr(col("v", 2)), q.CountAggregateExpr(lit_i64(1)), q.MinAggregateExpr(col("v", 2), I64),
            q.MaxAggregateExpr(col("p", 3), D), q.SumAggregateExpr(col("p", 3), D), q.AvgAggregateExpr(col("p", 3), D, pa.decimal128(19, 6)),
            q.SumAggregateExpr(col("f", 4), pa.float64())]


def test_sorted_run_aggregate(ctx, oracle, monkeypatch):
    """Round 4 (qk_agg_runs): an input whose equal keys are adjacent is aggregated as RUNS — no table, no compaction; the kernel
    checks the order itself. Forced here (QHIP_AGG_RUNS=2): runs of 1..7 rows over sizes around the thread / workgroup tiles, one-
    and multi-column keys (Int64 + Date32, a string), every cell kind, NULL values, NULL keys as a run of their own, ragged batches;
    then inputs that are NOT of that kind — unsorted keys, runs longer than the kernel folds — which must fall back to the hashed
    kernel with the same result and be remembered; then the automatic choice on the second execution of a many-group plan."""
    rng = np.random.default_rng(17)
    monkeypatch.setenv("QHIP_AGG_RUNS", "2")
    ctx.forget_plans()
    for n in (1, 3, 4, 5, 1023, 1024, 1025, 4097, 50_000):
        schema, batch = _runs_table(rng, n, 7)
        scan = table_scan(schema, [batch.slice(0, n // 3), batch.slice(n // 3)])
        for keys in ([col("k", 0)], [col("k", 0), col("d", 1)], [col("s", 5)]):
            plan = q.HashAggregate(None, scan, keys, _runs_aggs())
            _same(plan, oracle)
            assert ctx.last_stats()["main_kernel_name"] == "qk_agg_runs", (n, len(keys))
    # NULL keys: adjacent NULLs are one group (their key words are zero, the mask word tells them apart from a real zero). Sorted
    # NULLS LAST is "non-decreasing" for the kernel (the mask word leads the key); a NULL run in the MIDDLE is not — it falls back
    schema, batch = _runs_table(rng, 20_000, 5)
    kk = batch.column(0).to_numpy(zero_copy_only=False).astype(np.int64)
    for nullrun, runs in ((kk > kk[19_990], True), ((kk > kk[5000]) & (kk <= kk[5010]), False)):
        nb = pa.RecordBatch.from_arrays([pa.array(kk, type=I64, mask=nullrun)] + batch.columns[1:], schema=schema)
        ctx.forget_plans()
        plan = q.HashAggregate(None, table_scan(schema, [nb]), [col("k", 0)], _runs_aggs())
        _same(plan, oracle)
        assert (ctx.last_stats()["main_kernel_name"] == "qk_agg_runs") == runs
    # negative and positive keys in signed order, strings in bytewise order
    neg = pa.RecordBatch.from_arrays([pa.array(kk - kk[10_000], type=I64)] + batch.columns[1:], schema=schema)
    ctx.forget_plans()
    plan = q.HashAggregate(None, table_scan(schema, [neg]), [col("k", 0)], _runs_aggs()[:3])
    _same(plan, oracle)
    assert ctx.last_stats()["main_kernel_name"] == "qk_agg_runs"
    # runs longer than the kernel folds (QHIP_AGG_RUNS_MAX, default 256): flagged, hashed kernel, remembered
    schema, batch = _runs_table(rng, 30_000, 900)
    plan = q.HashAggregate(None, table_scan(schema, [batch]), [col("k", 0)], _runs_aggs()[:3])
    for _ in range(2):
        _same(plan, oracle)
        assert ctx.last_stats()["main_kernel_name"] != "qk_agg_runs"
    # unsorted keys
    ctx.forget_plans()
    schema, batch = _runs_table(rng, 30_000, 4)
    perm = rng.permutation(batch.num_rows)
    shuffled = batch.take(pa.array(perm))
    plan = q.HashAggregate(None, table_scan(schema, [shuffled]), [col("k", 0)], _runs_aggs())
    for _ in range(2):
        _same(plan, oracle)
        assert ctx.last_stats()["main_kernel_name"] != "qk_agg_runs"
    # ... one swap of two neighbouring runs is enough
    kk = batch.column(0).to_numpy(zero_copy_only=False).astype(np.int64).copy()
    a, b = kk[100], kk[20_000]
    kk[kk == a], kk[kk == b] = b, a
    swapped = pa.RecordBatch.from_arrays([pa.array(kk, type=I64)] + batch.columns[1:], schema=schema)
    ctx.forget_plans()
    plan = q.HashAggregate(None, table_scan(schema, [swapped]), [col("k", 0)], _runs_aggs()[:3])
    _same(plan, oracle)
    assert ctx.last_stats()["main_kernel_name"] != "qk_agg_runs"
    # the automatic choice: first execution through the table (nothing known), second as runs
    monkeypatch.delenv("QHIP_AGG_RUNS")
    ctx.forget_plans()
    schema, batch = _runs_table(rng, 120_000, 5)
    plan = q.HashAggregate(None, table_scan(schema, [batch]), [col("k", 0), col("d", 1)], _runs_aggs())
    names = []
    for _ in range(3):
        _same(plan, oracle)
        names.append(ctx.last_stats()["main_kernel_name"])
    assert names[0] != "qk_agg_runs" and names[1] == names[2] == "qk_agg_runs", names
    # a scan filter in the aggregate: no such entry point (a rejected row would split a run)
    pred = q.BinaryExpr(col("v", 2), Operator.Gt, lit_i64(0))
    monkeypatch.setenv("QHIP_AGG_RUNS", "2")
    plan = q.HashAggregate(None, table_scan(schema, [batch], pred), [col("k", 0)], _runs_aggs()[:3])
    _same(plan, oracle)
    assert ctx.last_stats()["main_kernel_name"] != "qk_agg_runs"
    ctx.forget_plans()


def test_sorted_run_aggregate_over_a_join_output(ctx, oracle, monkeypatch):
    """Q3's shape: the probe side is stored in key order, the join output keeps probe order, the aggregate above it groups by the
    probe key and columns of the build side — its input is a join output of deferred size read through index vectors."""
    rng = np.random.default_rng(23)
    n_orders, n_items = 30_000, 100_000
    os_ = pa.schema([pa.field("o_key", I64), pa.field("o_date", pa.date32()), pa.field("o_prio", pa.int32())])
    ls_ = pa.schema([pa.field("l_key", I64), pa.field("l_price", pa.decimal128(15, 2)), pa.field("l_disc", pa.decimal128(15, 2))])
    import decimal
    ob = pa.RecordBatch.from_arrays([pa.array(np.arange(n_orders) * 4 + 1, type=I64),
                                     pa.array(rng.integers(9000, 9400, n_orders).astype(np.int32), type=pa.int32()).cast(pa.date32()),
                                     pa.array(rng.integers(0, 3, n_orders), type=pa.int32())], schema=os_)
    lk = np.sort(rng.integers(0, n_orders, n_items)) * 4 + 1
    lb = pa.RecordBatch.from_arrays([pa.array(lk, type=I64),
                                     pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(100, 10**7, n_items)], type=pa.decimal128(15, 2)),
                                     pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(0, 11, n_items)], type=pa.decimal128(15, 2))], schema=ls_)
    opred = q.BinaryExpr(col("o_date", 1), Operator.Lt, q.CastExpr(q.Literal(q.ScalarValue.Int32(9200)), pa.date32()))
    join = q.HashJoinExec.try_new(table_scan(os_, [ob], opred), table_scan(ls_, [lb.slice(0, 40_000), lb.slice(40_000)]), JoinType.Inner,
                                  [(col("o_key", 0), col("l_key", 0))], None)
    rev = q.BinaryExpr(col("l_price", 4), Operator.Mul, col("l_disc", 5))
    plan = q.HashAggregate(None, join, [col("l_key", 3), col("o_date", 1), col("o_prio", 2)],
                           [q.SumAggregateExpr(rev, pa.decimal128(31, 4)), q.CountAggregateExpr(lit_i64(1))])
    names = []
    for _ in range(4):
        _same(plan, oracle)
        names.append(ctx.last_stats()["main_kernel_name"])
    assert names[-1] == "qk_agg_runs" and names[0] != "qk_agg_runs", names
