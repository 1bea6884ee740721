"""GPU parity of the aggregate kernel's NARROW arithmetic at its thresholds (csrc/codegen.cpp, csrc/relops.cpp
`ensure_value_bounds`): on inputs of >= 2^22 rows the kernel reads a cached per-column max |value| (rounded up to whole
bits) and multiplies 32 x 32 -> 64 when both operands are below 2^31, 64 x 64 -> 128 below 2^63, keeps arguments below
2^63 as i64 and accumulates SUMs of values below 2^39 in 64-bit lane accumulators. The reference computes all of it in
wrapping i128 (sum.rs:71-81, avg.rs:91-116, binary.rs:51-68 -> arrow's `*_wrapping`), so every choice must give the same
bits. Each case puts a column's extreme values exactly AT a threshold (2^31 - 1 | 2^31, 2^39 - 1 | 2^39, 2^63 - 1 | 2^63 and
their negatives, plus values beyond 64 bits), runs with the statistics (narrow code) and with QHIP_AGG_NO_BOUNDS=1 (the
128-bit code), and compares with EXACT Python integers reduced mod 2^128."""
import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator
from qurious_amd import ScalarValue as S

from .helpers import col, rows_of, table_scan

pytestmark = pytest.mark.gpu
N = (1 << 22) + 4321          # >= 2^22 rows: the value statistics engage (agg.cpp: ensure_value_bounds min_rows)
I64 = pa.int64()
M128 = 1 << 128


def _wrap(v: int) -> int:
    v %= M128
    return v - M128 if v >= (1 << 127) else v


def _dec_array(values_small: np.ndarray, extremes: dict, precision: int):
    """Decimal128(precision, 0) column: int64 `values_small` everywhere except rows `extremes` {row: python int} — built from
    the raw little-endian 16-byte words, so any 128-bit value goes in whatever the declared precision says"""
    lo = values_small.astype(np.int64).astype(np.uint64)
    hi = (values_small.astype(np.int64) >> 63).astype(np.uint64)          # sign extension
    for row, v in extremes.items():
        u = v % M128
        lo[row] = u & 0xFFFFFFFFFFFFFFFF
        hi[row] = u >> 64
    raw = np.empty((len(lo), 2), dtype=np.uint64)
    raw[:, 0], raw[:, 1] = lo, hi
    return pa.Array.from_buffers(pa.decimal128(precision, 0), len(lo), [None, pa.py_buffer(raw)])


def _i128_of(arr: pa.Array):
    """python ints of a Decimal128 array's raw values (no Decimal round trip: sums may exceed 38 digits, which arrow-rs's
    wrapping sum does not check either)"""
    buf = arr.buffers()[1]
    raw = np.frombuffer(buf, dtype=np.uint64, count=2 * (arr.offset + len(arr)))[2 * arr.offset:].reshape(-1, 2)
    valid = arr.is_valid().to_pylist()
    return [(_wrap(int(lo) | (int(hi) << 64)) if ok else None) for (lo, hi), ok in zip(raw.tolist(), valid)]


def _exact_sum(small: np.ndarray, extremes: dict, mask=None) -> int:
    sel = np.ones(len(small), dtype=bool) if mask is None else mask.copy()
    total = 0
    for row, v in extremes.items():
        if sel[row]:
            total += v
        sel[row] = False
    return total + int(small[sel].astype(object).sum()) if sel.any() else total


THRESHOLDS = [
    ("below 2^31", (1 << 31) - 1), ("at 2^31", 1 << 31), ("below 2^39", (1 << 39) - 1), ("at 2^39", 1 << 39),
    ("below 2^63", (1 << 63) - 1), ("at 2^63", 1 << 63), ("beyond 64 bits", (1 << 100) + 12345),
]


@pytest.mark.parametrize("name,extreme", THRESHOLDS)
@pytest.mark.parametrize("no_bounds", [False, True])
def test_sum_avg_and_products_at_the_narrow_arithmetic_thresholds(ctx, monkeypatch, name, extreme, no_bounds):
    if no_bounds:
        monkeypatch.setenv("QHIP_AGG_NO_BOUNDS", "1")
    else:
        monkeypatch.setenv("QHIP_NARROW_FIRST_USE", "1")   # (narrow copies at the first read: these tests execute a plan once)
    rng = np.random.default_rng(extreme % 1000 + (7 if no_bounds else 0))
    g = rng.integers(0, 4, N).astype(np.int64)
    a_small = rng.integers(-1000, 1000, N)
    b_small = rng.integers(-9, 10, N)
    rows = rng.permutation(N)[:64]
    # a: half the extreme rows at +extreme, half at -extreme (and -extreme - 1 where that still is the same bit count)
    a_ext = {int(r): (extreme if k % 2 == 0 else -extreme) for k, r in enumerate(rows[:32])}
    b_ext = {int(r): (extreme if k % 3 == 0 else -extreme) for k, r in enumerate(rows[16:48])}   # 16 rows carry BOTH extremes
    p = 38
    a, b = _dec_array(a_small, a_ext, p), _dec_array(b_small, b_ext, p)
    schema = pa.schema([pa.field("g", I64, False), pa.field("a", a.type, False), pa.field("b", b.type, False)])
    cuts = list(range(0, N, 1 << 20)) + [N]
    batches = [pa.RecordBatch.from_arrays([pa.array(g[s:e], I64), a.slice(s, e - s), b.slice(s, e - s)], schema=schema) for s, e in zip(cuts[:-1], cuts[1:])]
    scan = table_scan(schema, batches)
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))
    ab = q.BinaryExpr(col("a", 1), Operator.Mul, col("b", 2))
    a1b = q.BinaryExpr(col("a", 1), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("b", 2)))
    t = pa.decimal128(38, 0)
    aggs = [q.SumAggregateExpr(col("a", 1), t), q.SumAggregateExpr(ab, t), q.SumAggregateExpr(a1b, t), q.CountAggregateExpr(col("a", 1)),
            q.MinAggregateExpr(col("a", 1), t), q.MaxAggregateExpr(col("b", 2), t)]
    out_schema = pa.schema([pa.field("g", I64)] + [pa.field(f"c{k}", t if k != 3 else I64) for k in range(6)])
    plan = q.HashAggregate(out_schema, scan, [col("g", 0)], aggs)
    got = plan.execute()
    assert len(got) == 1 and got[0].num_rows == 4
    av = {r: v for r, v in a_ext.items()}
    bv = {r: v for r, v in b_ext.items()}
    a_obj = a_small.astype(object)
    b_obj = b_small.astype(object)
    for r, v in av.items():
        a_obj[r] = v
    for r, v in bv.items():
        b_obj[r] = v
    special = sorted(set(av) | set(bv))
    keys = got[0].column(0).to_pylist()
    sums = [_i128_of(got[0].column(k)) for k in (1, 2, 3)]
    counts = got[0].column(4).to_pylist()
    mins, maxs = _i128_of(got[0].column(5)), _i128_of(got[0].column(6))
    for grp in range(4):
        k = keys.index(grp)
        m = g == grp
        ms = m.copy()
        ms[special] = False
        base_a = int(a_small[ms].sum())
        base_ab = int((a_small[ms] * b_small[ms]).sum())                  # |a| < 1000, |b| < 10: exact in int64
        base_a1b = int((a_small[ms] * (1 - b_small[ms])).sum())
        sp = [r for r in special if m[r]]
        want_a = base_a + sum(a_obj[r] for r in sp)
        want_ab = base_ab + sum(a_obj[r] * b_obj[r] for r in sp)
        want_a1b = base_a1b + sum(a_obj[r] * (1 - b_obj[r]) for r in sp)
        assert sums[0][k] == _wrap(want_a), (name, grp, "SUM(a)")
        assert sums[1][k] == _wrap(want_ab), (name, grp, "SUM(a * b)")
        assert sums[2][k] == _wrap(want_a1b), (name, grp, "SUM(a * (1 - b))")
        assert counts[k] == int(m.sum())
        assert mins[k] == min([int(a_small[ms].min())] + [a_obj[r] for r in sp])
        assert maxs[k] == max([int(b_small[ms].max())] + [b_obj[r] for r in sp])
    # the ungrouped form (whole-kernel per-thread accumulators: the longest lane sums)
    plan0 = q.NoGroupingAggregate(pa.schema([pa.field("s", t), pa.field("p", t)]), scan, [aggs[0], aggs[1]])
    got0 = plan0.execute()[0]
    assert _i128_of(got0.column(0))[0] == _wrap(sum(int(v) for v in a_obj)), name
    assert _i128_of(got0.column(1))[0] == _wrap(sum(int(x) * int(y) for x, y in zip(a_obj[special], b_obj[special])) +
                                                int((np.delete(a_small, special) * np.delete(b_small, special)).sum())), name


@pytest.mark.parametrize("extreme", [(1 << 31) - 1, 1 << 31, (1 << 39) - 1, 1 << 39])
@pytest.mark.parametrize("no_bounds", [False, True])
def test_truncating_decimal_avg_at_the_thresholds(ctx, monkeypatch, extreme, no_bounds):
    """AVG(Decimal128(p, s)) = (sum * 10^4) div count, truncating towards zero, typed (p + 4, s + 4) (avg.rs:91-116) — with the
    SUM behind it accumulated in 64-bit lanes below 2^39 and in 128 bits from there"""
    if no_bounds:
        monkeypatch.setenv("QHIP_AGG_NO_BOUNDS", "1")
    else:
        monkeypatch.setenv("QHIP_NARROW_FIRST_USE", "1")   # (narrow copies at the first read: these tests execute a plan once)
    rng = np.random.default_rng(extreme % 997)
    # EVERY row big, in (+x, -x + d) pairs of one group: a group's SUM stays small enough for AVG's checked sum * 10^4
    # (Decimal128(19, 6), avg.rs:105-116) while the LANE sums grow — a lane always sees rows of one parity, i.e. one sign
    half = N // 2
    gh = rng.integers(0, 3, half).astype(np.int64)
    x = rng.integers(extreme - 1000, extreme, half, endpoint=True)
    x[rng.permutation(half)[:16]] = extreme
    d = rng.integers(0, 1000, half)
    g = np.empty(N, dtype=np.int64)
    v = np.zeros(N, dtype=np.int64)
    g[0:2 * half:2], g[1:2 * half:2] = gh, gh
    v[0:2 * half:2], v[1:2 * half:2] = x, -x + d
    if N % 2:
        g[-1], v[-1] = 0, -extreme
    dec = pa.decimal128(15, 2)
    arr = pa.Array.from_buffers(dec, N, _dec_array(v, {}, 15).buffers())
    schema = pa.schema([pa.field("g", I64, False), pa.field("v", dec, False)])
    cuts = list(range(0, N, 1 << 20)) + [N]
    batches = [pa.RecordBatch.from_arrays([pa.array(g[s:e], I64), arr.slice(s, e - s)], schema=schema) for s, e in zip(cuts[:-1], cuts[1:])]
    scan = table_scan(schema, batches)
    out_t = pa.decimal128(19, 6)
    plan = q.HashAggregate(pa.schema([pa.field("g", I64), pa.field("avg", out_t), pa.field("sum", dec)]), scan, [col("g", 0)],
                           [q.AvgAggregateExpr(col("v", 1), dec, out_t), q.SumAggregateExpr(col("v", 1), dec)])
    got = plan.execute()[0]
    keys = got.column(0).to_pylist()
    avgs, sums = _i128_of(got.column(1)), _i128_of(got.column(2))
    for grp in range(3):
        k = keys.index(grp)
        m = g == grp
        total = int(v[m].astype(object).sum())
        cnt = int(m.sum())
        num = total * 10_000
        want = abs(num) // cnt * (1 if num >= 0 else -1)      # truncation towards zero (i128 div_wrapping)
        assert sums[k] == total and avgs[k] == want, (extreme, grp)


def test_bounded_column_read_through_a_joins_index_vector(ctx, oracle, monkeypatch):
    """InputCol::indirect: after a hash join the aggregate reads a Decimal128 column of the probe table THROUGH the join's
    index vector; the column's value statistic (computed when the base table was first aggregated) travels with the deferred
    gather and selects the narrow code there too. Values sit at 2^31 - 1 / 2^31 in two runs of the same plan shape."""
    rng = np.random.default_rng(5)
    nb = 1000
    ls = pa.schema([pa.field("bk", I64, False), pa.field("bg", I64, False)])
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(nb), I64), pa.array(rng.integers(0, 5, nb), I64)], schema=ls)
    for extreme in ((1 << 31) - 1, 1 << 31, (1 << 39) - 1, 1 << 39):
        pk = rng.integers(0, nb + 50, N)
        x_small = rng.integers(-500, 500, N)
        y_small = rng.integers(0, 11, N)
        rows = rng.permutation(N)[:40]
        x_ext = {int(r): (extreme if k % 2 else -extreme) for k, r in enumerate(rows)}
        y_ext = {int(r): extreme for r in rows[:20]}
        x, y = _dec_array(x_small, x_ext, 38), _dec_array(y_small, y_ext, 38)
        rs = pa.schema([pa.field("pk", I64, False), pa.field("x", x.type, False), pa.field("y", y.type, False)])
        cuts = list(range(0, N, 1 << 20)) + [N]
        batches = [pa.RecordBatch.from_arrays([pa.array(pk[s:e], I64), x.slice(s, e - s), y.slice(s, e - s)], schema=rs) for s, e in zip(cuts[:-1], cuts[1:])]
        probe = table_scan(rs, batches)
        t = pa.decimal128(38, 0)
        xy = q.BinaryExpr(col("x", 1), Operator.Mul, col("y", 2))
        # (1) aggregate the base table: the statistics of x and y are computed and cached on its columns
        base = q.NoGroupingAggregate(pa.schema([pa.field("s", t)]), probe, [q.SumAggregateExpr(xy, t)])
        x_obj, y_obj = x_small.astype(object), y_small.astype(object)
        for r, v in x_ext.items():
            x_obj[r] = v
        for r, v in y_ext.items():
            y_obj[r] = v
        sp = sorted(set(x_ext) | set(y_ext))
        keep = np.ones(N, dtype=bool)
        keep[sp] = False
        assert _i128_of(base.execute()[0].column(0))[0] == _wrap(int((x_small[keep] * y_small[keep]).sum()) + sum(x_obj[r] * y_obj[r] for r in sp))
        # (2) join, then aggregate x * y grouped by a build-side column, reading x / y through the join's index vector
        for no_bounds in (False, True):
            if no_bounds:
                monkeypatch.setenv("QHIP_AGG_NO_BOUNDS", "1")
            else:
                monkeypatch.delenv("QHIP_AGG_NO_BOUNDS", raising=False)
            join = q.HashJoinExec.try_new(table_scan(ls, [lb]), probe, JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
            xy_j = q.BinaryExpr(col("x", 3), Operator.Mul, col("y", 4))
            agg = q.HashAggregate(pa.schema([pa.field("bg", I64), pa.field("s", t), pa.field("n", I64)]), join, [col("bg", 1)],
                                  [q.SumAggregateExpr(xy_j, t), q.CountAggregateExpr(q.Literal(S.Int64(1)))])
            for _ in range(2):   # (the second execution runs the join with its size left on the device)
                got = agg.execute()[0]
                keys = got.column(0).to_pylist()
                sums, counts = _i128_of(got.column(1)), got.column(2).to_pylist()
                bg = lb.column(1).to_numpy()
                matched = pk < nb
                grp_of_row = np.where(matched, bg[np.minimum(pk, nb - 1)], -1)
                for grp in range(5):
                    k = keys.index(grp)
                    m = grp_of_row == grp
                    mk = m & keep
                    want = int((x_small[mk] * y_small[mk]).sum()) + sum(x_obj[r] * y_obj[r] for r in sp if m[r])
                    assert sums[k] == _wrap(want) and counts[k] == int(m.sum()), (extreme, no_bounds, grp)


@pytest.mark.parametrize("extreme,width", [((1 << 31) - 1, 4), (1 << 31, 8), ((1 << 63) - 1, 8), (1 << 63, 16)])
def test_narrow_copies_of_decimal_columns(ctx, monkeypatch, extreme, width):
    """A Decimal128 column whose every value fits 32 / 64 bits gets a 4- / 8-byte copy that the aggregate kernel streams
    instead of the 16-byte values (common.hpp DevColumn::narrow, relops.cpp ensure_value_bounds). Checked: the copy's width at
    the thresholds (from the bytes per row the kernel reports), negative values and NULL slots, the same sums as with
    QHIP_NARROW_DECIMALS=0 and as exact Python integers — and that the table's 16-byte values still serve every other operator
    afterwards (a Filter over the same device table returns the original values)."""
    rng = np.random.default_rng(extreme % 977)
    g = rng.integers(0, 3, N).astype(np.int64)
    small = rng.integers(-50000, 50000, N)
    rows = rng.permutation(N)[:40]
    ext = {int(r): (extreme if k % 2 == 0 else -extreme) for k, r in enumerate(rows)}
    a = _dec_array(small, ext, 38)
    valid = rng.random(N) > 0.01
    valid[rows] = True
    a_nulls = pa.Array.from_buffers(a.type, N, [pa.array(valid).buffers()[1], a.buffers()[1]])
    schema = pa.schema([pa.field("g", I64, False), pa.field("a", a.type, True)])
    cuts = list(range(0, N, 1 << 20)) + [N]
    batches = [pa.RecordBatch.from_arrays([pa.array(g[s:e], I64), a_nulls.slice(s, e - s)], schema=schema) for s, e in zip(cuts[:-1], cuts[1:])]
    scan = table_scan(schema, batches)
    t = pa.decimal128(38, 0)
    out_schema = pa.schema([pa.field("g", I64), pa.field("s", t), pa.field("c", I64)])
    plan = q.HashAggregate(out_schema, scan, [col("g", 0)], [q.SumAggregateExpr(col("a", 1), t), q.CountAggregateExpr(col("a", 1))])

    def run():
        out = plan.execute_device()
        st = ctx.last_stats()
        b = out.to_batches()[0]
        return sorted(zip(b.column(0).to_pylist(), _i128_of(b.column(1)), b.column(2).to_pylist())), st

    first, st_first = run()
    # the first big read collects the column's statistics; the same pass writes the 4-byte copy speculatively (round 4), so a
    # column that fits 31 bits is streamed narrow from the first execution on; the 8-byte copy is made at the second big read
    assert abs(st_first["bytes_per_row_read"] - (8 + (4 if width == 4 else 16) + 0.125)) < 1e-9, st_first["bytes_per_row_read"]
    got, st = run()
    assert got == first
    # 8 bytes of group key + the value column at its narrow width + its validity bits
    assert abs(st["bytes_per_row_read"] - (8 + width + 0.125)) < 1e-9, st["bytes_per_row_read"]
    a_obj = small.astype(object)
    for r, v in ext.items():
        a_obj[r] = v
    for grp, s, c in got:
        m = (g == grp) & valid
        assert c == int(m.sum())
        assert s == _wrap(sum(int(v) for v in a_obj[m])), (grp, width)
    monkeypatch.setenv("QHIP_NARROW_DECIMALS", "0")
    ctx.forget_plans()
    wide, st_wide = run()
    assert wide == got and abs(st_wide["bytes_per_row_read"] - (8 + 16 + 0.125)) < 1e-9
    monkeypatch.delenv("QHIP_NARROW_DECIMALS")
    # the canonical 16-byte values are untouched: a Filter over the same device table exports the original extremes
    keep = q.Filter(scan, q.BinaryExpr(col("a", 1), Operator.Gt, q.CastExpr(q.Literal(S.Int64(60000)), a.type)))
    kept = [v for b in keep.execute() for v in _i128_of(b.column(1))]
    assert sorted(kept) == sorted(v for r, v in ext.items() if v > 60000)


def test_narrow_copy_for_columns_only_ever_read_through_an_index_vector(ctx, oracle, monkeypatch):
    """Round 4: Q3's aggregate reads lineitem's prices and discounts only through the join's index vector — the table's own
    column objects are never streamed, so they never got statistics or narrow copies. Those now live in the object all copies of
    a column share (ColRange) and are made at the second such read: the third execution gathers 4-byte values (and multiplies in
    64 bits) instead of 16-byte ones. Results must not change; columns whose values do not fit 31 bits keep their 16 bytes."""
    import decimal
    monkeypatch.setenv("QHIP_STATS_MIN_ROWS", "1")
    rng = np.random.default_rng(11)
    nb, npr = 3000, 200_000
    D = pa.decimal128(15, 2)
    ls = pa.schema([pa.field("bk", I64, False), pa.field("bg", pa.int32(), False)])
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(nb) * 2, I64), pa.array(rng.integers(0, 40, nb), pa.int32())], schema=ls)
    dec = lambda v: pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in v], type=D)   # noqa: E731
    big = rng.integers(-10**6, 10**6, npr).astype(object)
    big[7] = 1 << 40                                                                     # does not fit 31 bits
    rs = pa.schema([pa.field("pk", I64, False), pa.field("price", D, False), pa.field("disc", D, False), pa.field("wide", D, False)])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2 * nb, npr), I64), dec(rng.integers(100, 10**7, npr)), dec(rng.integers(0, 11, npr)), dec(big)], schema=rs)
    join = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb.slice(0, 70_000), rb.slice(70_000)]), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))                        # what the reference's planner makes of `1 - l_discount`
    rev = q.BinaryExpr(col("price", 3), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("disc", 4)))   # Decimal128(38, 4)
    plan = q.HashAggregate(None, join, [col("bg", 1)], [q.SumAggregateExpr(rev, pa.decimal128(38, 4)),
                                                        q.SumAggregateExpr(col("wide", 5), D), q.MinAggregateExpr(col("price", 3), D)])
    want = sorted(map(repr, rows_of(oracle.execute(plan))))
    read = []
    for _ in range(4):
        assert sorted(map(repr, rows_of(plan.execute()))) == want
        read.append(ctx.last_stats()["bytes_per_row_read"])
    assert read[-1] < read[0] - 20, read          # price and disc: 16 -> 4 bytes each; wide stays 16


@pytest.mark.gpu
def test_record_copies_for_columns_read_through_one_index_vector(ctx, oracle, monkeypatch):
    """Round 4: an aggregate over a join output reads each column as source[index[row]] — one random 64-byte access per row and
    COLUMN. Columns of one source table behind one index vector (Q3: price | discount of lineitem, date | priority of orders) are
    interleaved into 8- / 16-byte records at the second such read (ColRange::rec_buf, relops.cpp ensure_indirect_records) and
    the generated kernel reads fields of those. Results must not change from execution to execution; the copies show up in the
    tables' auxiliary bytes; a column of 16 bytes stays out; the switch turns it off."""
    import decimal
    monkeypatch.setenv("QHIP_STATS_MIN_ROWS", "1")
    monkeypatch.setenv("QHIP_INDIRECT_RECORDS", "1")
    rng = np.random.default_rng(21)
    nb, npr = 4000, 150_000
    D = pa.decimal128(15, 2)
    dec = lambda v: pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in v], type=D)   # noqa: E731
    ls = pa.schema([pa.field("bk", I64, False), pa.field("g1", pa.int32(), False), pa.field("g2", pa.date32(), False), pa.field("g3", I64, False)])
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(nb) * 2, I64), pa.array(rng.integers(0, 30, nb), pa.int32()),
                                     pa.array(rng.integers(9000, 9004, nb), pa.int32()).cast(pa.date32()), pa.array(rng.integers(-10**12, 10**12, nb), I64)], schema=ls)
    big = rng.integers(-10**6, 10**6, npr).astype(object)
    big[3] = 1 << 40
    rs = pa.schema([pa.field("pk", I64, False), pa.field("price", D, False), pa.field("disc", D, False), pa.field("qty", pa.int32(), False), pa.field("wide", D, False)])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2 * nb, npr), I64), dec(rng.integers(100, 10**7, npr)), dec(rng.integers(0, 11, npr)),
                                     pa.array(rng.integers(1, 51, npr), pa.int32()), dec(big)], schema=rs)
    lscan, rscan = table_scan(ls, [lb]), table_scan(rs, [rb.slice(0, 50_000), rb.slice(50_000)])
    join = q.HashJoinExec.try_new(lscan, rscan, JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    one = q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0))
    rev = q.BinaryExpr(col("price", 5), Operator.Mul, q.BinaryExpr(one, Operator.Sub, col("disc", 6)))
    plan = q.HashAggregate(None, join, [col("g1", 1), col("g2", 2)],
                           [q.SumAggregateExpr(rev, pa.decimal128(38, 4)), q.SumAggregateExpr(col("g3", 3), I64), q.MinAggregateExpr(col("qty", 7), pa.int32()),
                            q.SumAggregateExpr(col("wide", 8), D), q.CountAggregateExpr(q.Literal(S.Int64(1)))])
    want = sorted(map(repr, rows_of(oracle.execute(plan))))
    aux = []
    for _ in range(6):
        assert sorted(map(repr, rows_of(plan.execute()))) == want
        aux.append((lscan.datasource.device_table().aux_bytes, rscan.datasource.device_table().aux_bytes))
    # build side: g1 | g2 | g3 = 4 + 4 + 8 bytes -> 16-byte records; probe side: price | disc (narrow, 4 + 4) | qty (4) -> 16-byte
    # records beside the two narrow copies (`wide` has no narrow copy and stays a 16-byte gather)
    assert aux[-1][0] >= nb * 16 and aux[0][0] == 0, aux
    assert aux[-1][1] >= npr * (16 + 4 + 4) and aux[-1][1] > aux[1][1], aux
    # a second plan over the same tables that reads ONE of those columns per side uses the records that exist
    plan2 = q.HashAggregate(None, join, [col("g1", 1)], [q.MaxAggregateExpr(col("qty", 7), pa.int32())])
    for _ in range(2):
        assert sorted(map(repr, rows_of(plan2.execute()))) == sorted(map(repr, rows_of(oracle.execute(plan2))))
    assert (lscan.datasource.device_table().aux_bytes, rscan.datasource.device_table().aux_bytes) == aux[-1]
    # ... and with the switch off nothing is made
    monkeypatch.setenv("QHIP_INDIRECT_RECORDS", "0")
    l2, r2 = table_scan(ls, [lb]), table_scan(rs, [rb])
    j2 = q.HashJoinExec.try_new(l2, r2, JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    p3 = q.HashAggregate(None, j2, plan.group_exprs, plan.aggregate_exprs)
    for _ in range(4):
        assert sorted(map(repr, rows_of(p3.execute()))) == want
    assert l2.datasource.device_table().aux_bytes == 0
