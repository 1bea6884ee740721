"""GPU parity tests of the exchange's fused filter + partition (qhip_partition_filtered): the part of every row
against the numpy mirror (oracle/qoracle.py partition_ids), the rows of every part against an Arrow filter of the input in
input order, for every column kind (fixed width 1..16 bytes, validity bitmaps, Utf8, Boolean, NULL-typed placeholders),
every ranking variant of pass 2 (<= 8, <= 16, up to 255 parts), scan filters with NULLs, dropped columns, join outputs
whose columns are still deferred gathers, ragged and empty inputs. The reference has no exchange operator (single process):
what is pinned is the partition function both join sides and all ranks must agree on, and that no row is lost or reordered."""
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import qurious_amd as q
from qurious_amd import exchange
from qurious_amd.datatypes import JoinType, Operator

from .helpers import col, lit_i64, rows_of, table_scan

pytestmark = pytest.mark.gpu
I64 = pa.int64()


def _mixed_batch(n, seed, null_frac=0.03):
    rng = np.random.default_rng(seed)
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("d", pa.decimal128(15, 2)), pa.field("b", pa.bool_()),
                        pa.field("i", pa.int32()), pa.field("u", pa.uint8()), pa.field("f", pa.float64()), pa.field("t", pa.date32()),
                        pa.field("h", pa.int16())])
    mask = lambda: rng.random(n) < null_frac   # noqa: E731
    batch = pa.RecordBatch.from_arrays([
        pa.array(rng.integers(-50, 5000, n), type=I64, mask=mask()),
        pa.array(["v%d" % v if v % 11 else "" for v in rng.integers(0, 300, n)], type=pa.string(), mask=mask()),
        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**8, 10**8, n)], type=pa.decimal128(15, 2)),
        pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=mask()),
        pa.array(rng.integers(-1000, 1000, n), type=pa.int32()),
        pa.array(rng.integers(0, 255, n), type=pa.uint8()),
        pa.array(rng.normal(size=n), type=pa.float64(), mask=mask()),
        pa.array(rng.integers(8000, 10500, n).astype(np.int32), type=pa.int32()).cast(pa.date32()),
        pa.array(rng.integers(-30000, 30000, n), type=pa.int16())], schema=schema)
    return schema, batch


def _check_parts(parts, batch, pid, pass_mask=None, keep=None):
    n_parts = len(parts)
    for p, part in enumerate(parts):
        m = pid == p
        if pass_mask is not None:
            m = m & pass_mask
        want_b = batch.filter(pa.array(m))
        got_b = part.to_batches()
        got_rows = rows_of(got_b)
        want_rows = rows_of([want_b])
        if keep is not None:
            want_rows = [tuple(v if keep[c] else None for c, v in enumerate(r)) for r in want_rows]
        assert part.num_rows == want_b.num_rows, (p, n_parts)
        assert [repr(r) for r in got_rows] == [repr(r) for r in want_rows], (p, n_parts)   # repr: NaN == NaN


@pytest.mark.parametrize("n_parts", [1, 2, 3, 8, 9, 16, 17, 100, 255])
def test_every_ranking_variant_keeps_rows_and_order(ctx, oracle, n_parts):
    schema, batch = _mixed_batch(30_011, 5 + n_parts)
    dev = table_scan(schema, [batch.slice(0, 7), batch.slice(7, 12_000), batch.slice(12_007)]).execute_device()
    parts = exchange.partition_filtered(dev, [col("k", 0)], n_parts)
    pid = oracle.partition_ids([batch.column("k")], n_parts)
    _check_parts(parts, batch, pid)
    assert sum(p.num_rows for p in parts) == batch.num_rows


def test_scan_filter_and_dropped_columns(ctx, oracle):
    schema, batch = _mixed_batch(70_001, 99, null_frac=0.1)
    dev = table_scan(schema, [batch]).execute_device()
    # i > -200 AND k < 4000: NULL k -> NULL predicate -> the row is dropped (filter_record_batch semantics)
    pred = q.BinaryExpr(q.BinaryExpr(col("i", 4), Operator.Gt, q.Literal(q.ScalarValue.Int32(-200))), Operator.And,
                        q.BinaryExpr(col("k", 0), Operator.Lt, lit_i64(4000)))
    keep = [True, False, True, True, False, True, False, True, False]
    parts = exchange.partition_filtered(dev, [col("t", 7), col("u", 5)], 8, predicate=pred, keep=keep)
    pid = oracle.partition_ids([batch.column("t"), batch.column("u")], 8)
    passes = np.array(pc.fill_null(pc.and_kleene(pc.greater(batch.column("i"), -200), pc.less(batch.column("k"), 4000)), False))
    _check_parts(parts, batch, pid, passes, keep)
    for part in parts:   # dropped columns are NULL-typed placeholders (what qhip_table_keep_columns makes of them)
        b = part.to_batches()[0] if part.num_rows else None
        if b is not None:
            assert [str(f.type) for f in b.schema][1] == "null" and [str(f.type) for f in b.schema][4] == "null"


def test_nullable_and_utf8_and_decimal_keys(ctx, oracle):
    schema, batch = _mixed_batch(20_000, 3, null_frac=0.2)
    dev = table_scan(schema, [batch]).execute_device()
    for keys, arrays, n_parts in (([col("s", 1)], ["s"], 5), ([col("d", 2), col("k", 0)], ["d", "k"], 8), ([col("k", 0), col("s", 1)], ["k", "s"], 12)):
        parts = exchange.partition_filtered(dev, keys, n_parts)
        pid = oracle.partition_ids([batch.column(a) for a in arrays], n_parts)
        _check_parts(parts, batch, pid)


def test_empty_tiny_and_all_rejected_inputs(ctx, oracle):
    schema, batch = _mixed_batch(300, 8)
    for rows in (0, 1, 63, 64, 65, 255, 256, 257):
        b = batch.slice(0, rows)
        dev = table_scan(schema, [b]).execute_device()
        parts = exchange.partition_filtered(dev, [col("k", 0)], 4)
        _check_parts(parts, b, oracle.partition_ids([b.column("k")], 4) if rows else np.zeros(0, dtype=np.int64))
    dev = table_scan(schema, [batch]).execute_device()
    none = exchange.partition_filtered(dev, [col("k", 0)], 8, predicate=q.BinaryExpr(col("i", 4), Operator.Gt, q.Literal(q.ScalarValue.Int32(5000))))
    assert [p.num_rows for p in none] == [0] * 8
    assert exchange.concat_tables(none).num_rows == 0


def test_join_output_with_deferred_columns_is_partitioned_through_its_index_vectors(ctx, oracle):
    rng = np.random.default_rng(12)
    ls = pa.schema([pa.field("a", I64), pa.field("x", pa.int32()), pa.field("ls", pa.string())])
    rs = pa.schema([pa.field("b", I64), pa.field("y", pa.decimal128(15, 2))])
    nl, nr = 5_000, 40_000
    left = pa.RecordBatch.from_arrays([pa.array(np.arange(nl), type=I64), pa.array(rng.integers(0, 100, nl), type=pa.int32()),
                                       pa.array(["s%d" % (v % 17) for v in range(nl)])], schema=ls)
    right = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2 * nl, nr), type=I64),
                                        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(0, 10**6, nr)], type=pa.decimal128(15, 2))], schema=rs)
    join = q.HashJoinExec.try_new(table_scan(ls, [left]), table_scan(rs, [right]), JoinType.Inner, [(col("a", 0), col("b", 0))])
    out = join.execute_device()           # columns: deferred gathers over both sides
    want = join.execute()
    whole = pa.Table.from_batches(want).combine_chunks().to_batches()[0]
    keep = [True, True, False, False, True]
    parts = exchange.partition_filtered(out, [col("a", 0)], 8, keep=keep)
    pid = oracle.partition_ids([whole.column(0)], 8)
    _check_parts(parts, whole, pid, None, keep)
    # ... and with a string column kept (gathered per part through the selection vector)
    parts = exchange.partition_filtered(out, [col("b", 3)], 3, predicate=q.BinaryExpr(col("x", 1), Operator.Lt, q.Literal(q.ScalarValue.Int32(50))))
    pid = oracle.partition_ids([whole.column(3)], 3)
    _check_parts(parts, whole, pid, np.array(pc.less(whole.column(1), 50)))


def test_parts_round_trip_through_wire_images_and_concat(ctx, oracle):
    schema, batch = _mixed_batch(25_000, 21)
    dev = table_scan(schema, [batch]).execute_device()
    parts = exchange.partition_filtered(dev, [col("k", 0)], 8)
    packed = [exchange.pack_table(p) for p in parts]
    back = exchange.unpack_concat(ctx, schema, [m for m, _ in packed], [img for _, img in packed])
    assert sorted(map(repr, rows_of(back.to_batches()))) == sorted(map(repr, rows_of([batch])))
    assert [repr(r) for r in rows_of(exchange.concat_tables(parts).to_batches())] == [repr(r) for p in parts for r in rows_of(p.to_batches())]


def test_the_generic_path_still_answers_the_same(ctx, oracle, monkeypatch):
    schema, batch = _mixed_batch(9_000, 4)
    dev = table_scan(schema, [batch]).execute_device()
    fused = exchange.partition_filtered(dev, [col("k", 0)], 6)
    monkeypatch.setenv("QHIP_PARTITION_FUSED", "0")
    generic = exchange.partition_filtered(dev, [col("k", 0)], 6)
    for a, b in zip(fused, generic):
        assert [repr(r) for r in rows_of(a.to_batches())] == [repr(r) for r in rows_of(b.to_batches())]
    with pytest.raises(q.UnsupportedError):
        exchange.partition_filtered(dev, [col("k", 0)], 6, predicate=q.BinaryExpr(col("k", 0), Operator.Lt, lit_i64(1)))


def test_many_rows_per_unit_and_many_units(ctx, oracle, monkeypatch):
    """row ranges of one tile per wavefront (many units, a multi-launch scan) and of the whole table (one unit)"""
    schema, batch = _mixed_batch(200_003, 6, null_frac=0.0)
    dev = table_scan(schema, [batch]).execute_device()
    pid = oracle.partition_ids([batch.column("k")], 8)
    for rpu in ("256", "1048576"):
        monkeypatch.setenv("QHIP_PART_ROWS_PER_UNIT", rpu)
        parts = exchange.partition_filtered(dev, [col("k", 0)], 8, keep=[True, False, True, False, True, True, False, True, True])
        _check_parts(parts, batch, pid, None, [True, False, True, False, True, True, False, True, True])


def test_partition_by_key_range(ctx, oracle):
    """Round 4 (qhip_partition_filtered_by_range): the part of a row chosen by the RANGE its key falls in — bounds that cut between,
    exactly at and outside the key values, equal bounds (an empty part), negative keys, a Date32 key, NULL keys (they go where 0
    goes), a scan filter and dropped columns alongside, 2 .. 17 parts; every part keeps the input's row order. And what the
    exchange relies on: two tables split by the same bounds are co-located key by key."""
    schema, batch = _mixed_batch(60_007, 77, null_frac=0.05)
    dev = table_scan(schema, [batch.slice(0, 20_000), batch.slice(20_000)]).execute_device()
    for bounds in ([1000], [-60, 0, 4999], [-50, -50, 10, 10, 2500, 7000], list(range(-40, 4800, 300))):
        parts = exchange.partition_filtered(dev, [col("k", 0)], len(bounds) + 1, range_bounds=bounds)
        pid = oracle.partition_ids_by_range(batch.column("k"), bounds)
        _check_parts(parts, batch, pid)
        assert sum(p.num_rows for p in parts) == batch.num_rows
    pred = q.BinaryExpr(col("i", 4), Operator.Gt, q.Literal(q.ScalarValue.Int32(0)))
    keep = [True, False, True, False, True, True, False, True, False]
    bounds = [8500, 9000, 9900]
    parts = exchange.partition_filtered(dev, [col("t", 7)], 4, predicate=pred, keep=keep, range_bounds=bounds)
    passes = np.array(pc.fill_null(pc.greater(batch.column("i"), 0), False))
    _check_parts(parts, batch, oracle.partition_ids_by_range(batch.column("t"), bounds), passes, keep)
    with pytest.raises(q.UnsupportedError):
        exchange.partition_filtered(dev, [col("s", 1)], 3, range_bounds=[1, 2])           # not an integer-like key
    with pytest.raises(Exception):
        exchange.partition_filtered(dev, [col("k", 0)], 3, range_bounds=[5, 1])           # bounds not ascending
    assert exchange.column_range(dev, 0)[0] <= -50 and exchange.column_range(dev, 0)[1] >= 4999
