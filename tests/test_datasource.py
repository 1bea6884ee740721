"""File-backed tables with lazy per-column upload (datasource/file/{csv,parquet,json}.rs)."""
from __future__ import annotations

import decimal

import numpy as np
import pyarrow as pa
import pyarrow.csv as pacsv
import pyarrow.parquet as pq
import pytest

import qurious_amd as q
from oracle import qoracle
from qurious_amd import Operator
from qurious_amd import ScalarValue as S

from .helpers import col, rows_of

D = decimal.Decimal


def _wide_table(n=5000, seed=5):
    rng = np.random.default_rng(seed)
    dec = pa.decimal128(15, 2)
    cols = {
        "k": pa.array(rng.integers(0, 7, n), type=pa.int64()),
        "flag": pa.array([("A", "N", "R")[v] for v in rng.integers(0, 3, n)], type=pa.string()),
        "qty": pa.array([D(int(v)).scaleb(-2) for v in rng.integers(100, 5100, n)], type=dec),
        "price": pa.array(rng.random(n) * 1000.0, type=pa.float64()),
        "day": pa.array(rng.integers(9000, 9400, n), type=pa.int32()).cast(pa.date32()),
        "comment": pa.array(["comment number %d with some padding" % v for v in rng.integers(0, 10**6, n)], type=pa.string()),
        "unused1": pa.array(rng.integers(0, 10**9, n), type=pa.int64()),
        "unused2": pa.array(rng.random(n), type=pa.float64(), mask=rng.random(n) < 0.2),
    }
    return pa.table(cols)


def test_readers_build_lazy_memory_tables(tmp_path):
    tbl = _wide_table(1000)
    pq.write_table(tbl, tmp_path / "t.parquet")
    t = q.read_parquet(str(tmp_path / "t.parquet"), batch_rows=256)
    assert t.lazy_upload and t.schema() == tbl.schema and [b.num_rows for b in t.data] == [256, 256, 256, 232]
    assert pa.Table.from_batches(t.data).equals(tbl)
    sub = q.read_parquet(str(tmp_path / "t.parquet"), columns=["k", "qty"])
    assert [f.name for f in sub.schema()] == ["k", "qty"]
    csv_src = tbl.select(["k", "flag", "price"])
    pacsv.write_csv(csv_src, tmp_path / "t.csv")
    t = q.read_csv(str(tmp_path / "t.csv"))
    assert [f.name for f in t.schema()] == ["k", "flag", "price"] and t.schema().field(0).type == pa.int64()
    pacsv.write_csv(csv_src, tmp_path / "nohdr.csv", write_options=pacsv.WriteOptions(include_header=False, delimiter="|"))
    schema = pa.schema([pa.field("k", pa.int32()), pa.field("flag", pa.string()), pa.field("price", pa.float64())])
    t = q.read_csv(str(tmp_path / "nohdr.csv"), q.CsvReadOptions(has_header=False, delimiter="|"), schema=schema)
    assert t.schema() == schema and sum(b.num_rows for b in t.data) == 1000
    (tmp_path / "t.json").write_text('{"a": 1, "b": "x"}\n{"a": 2, "b": null}\n')
    assert rows_of(q.read_json(str(tmp_path / "t.json")).data) == [(1, "x"), (2, None)]


@pytest.mark.gpu
def test_gpu_lazy_upload_moves_only_the_columns_a_query_reads(tmp_path):
    q.get_context()
    tbl = _wide_table()
    pq.write_table(tbl, tmp_path / "t.parquet")
    table = q.read_parquet(str(tmp_path / "t.parquet"), batch_rows=1024)
    scan = q.Scan(table.schema(), table, None, q.BinaryExpr(col("day", 4), Operator.Lt, q.Literal(S.Date32(9300))))
    agg = q.HashAggregate(None, scan, [col("flag", 1)], [q.SumAggregateExpr(col("qty", 2), pa.decimal128(15, 2)), q.CountAggregateExpr(col("k", 0))])
    got = sorted(rows_of(agg.execute()))
    dev = table.device_table()
    resident = [dev.column_bytes(c) for c in range(dev.num_columns)]
    assert all(resident[c] > 0 for c in (0, 1, 2, 4)) and all(resident[c] == 0 for c in (3, 5, 6, 7)), resident
    eager = q.MemoryTable.try_new(table.schema(), table.data)
    want_plan = q.HashAggregate(None, q.Scan(eager.schema(), eager, None, scan.filter), agg.group_exprs, agg.aggregate_exprs)
    assert got == sorted(rows_of(qoracle.execute(want_plan))) == sorted(rows_of(want_plan.execute()))
    # operators that pass columns through keep them un-uploaded; an export uploads what it exports
    lim = q.Limit(q.Scan(table.schema(), table, None, None), 10, 5)
    out = lim.execute()
    assert rows_of(out) == rows_of(qoracle.execute(lim))
    assert [dev.column_bytes(c) > 0 for c in range(dev.num_columns)] == [True] * 8
    # a join / sort over a lazily uploaded table
    srt = q.Sort([q.PhysicalSortExpr(col("price", 3), q.SortOptions(True, True))], q.Scan(table.schema(), q.read_parquet(str(tmp_path / "t.parquet")), None, None), 7)
    assert rows_of(srt.execute()) == rows_of(qoracle.execute(srt))
