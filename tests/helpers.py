"""Helpers shared by the parity tests: fixture builders in the style of the reference's test_utils.rs."""
from __future__ import annotations

import decimal
from typing import Dict, List, Sequence

import pyarrow as pa

import qurious_amd as q


def build_table_scan_i32(columns: Dict[str, Sequence]) -> q.Scan:
    """test_utils.rs:218-236 build_table_scan_i32: nullable Int32 columns, one batch, MemoryTable -> Scan."""
    fields = [pa.field(name, pa.int32(), True) for name in columns]
    schema = pa.schema(fields)
    batch = pa.RecordBatch.from_arrays([pa.array(list(v), type=pa.int32()) for v in columns.values()], schema=schema)
    return q.Scan(schema, q.MemoryTable.try_new(schema, [batch]), None, None)


def table_scan(schema: pa.Schema, rows_or_batches, filter=None) -> q.Scan:
    """MemoryTable -> Scan over explicit batches (list of RecordBatch) or one batch built from row tuples."""
    if rows_or_batches and isinstance(rows_or_batches[0], pa.RecordBatch):
        batches = list(rows_or_batches)
    else:
        cols = list(zip(*rows_or_batches)) if rows_or_batches else [[] for _ in schema]
        batches = [pa.RecordBatch.from_arrays([pa.array(list(c), type=f.type) for c, f in zip(cols, schema)], schema=schema)]
    return q.Scan(schema, q.MemoryTable.try_new(schema, batches), None, filter)


def rows_of(batches: List[pa.RecordBatch]) -> List[tuple]:
    """All rows of a Vec<RecordBatch> as python tuples (Decimal -> unscaled-preserving decimal.Decimal)."""
    out = []
    for b in batches:
        cols = [c.to_pylist() for c in b.columns]
        out.extend(zip(*cols) if cols else [])
    return [tuple(r) for r in out]


def sort_key(row):
    return tuple((0, "") if v is None else (1, str(type(v).__name__), v) if not isinstance(v, (int, float, decimal.Decimal)) else (1, "n", v) for v in row)


def sorted_rows(batches):
    return sorted(rows_of(batches), key=lambda r: tuple((v is not None, v if v is not None else 0) if not isinstance(v, str) else (True, v) for v in r))


def col(name: str, index: int) -> q.Column:
    return q.Column(name, index)


def lit_i64(v):
    return q.Literal(q.ScalarValue.Int64(v))


def lit_i32(v):
    return q.Literal(q.ScalarValue.Int32(v))
