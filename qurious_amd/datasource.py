"""File-backed tables (reference: datasource/file/{csv,parquet,json}.rs + provider/table.rs:32-41).

The reference's readers load the whole file into a host `MemoryTable` with arrow-rs' readers and every scan hands every
column to the operators (`Scan.projections` is always None, planner/mod.rs:251-256). Here the file is decoded on the host
with Arrow C++ (pyarrow) into the same `MemoryTable`, whose device copy is LAZY: a column is uploaded to HBM the first time
an operator or an export reads it (qhip_table_from_arrow_lazy), so the columns a query never touches never cross PCIe —
for TPC-H Q1 that is 7 of lineitem's 16 columns. Parquet additionally skips the untouched column chunks on disk when a
`columns=` subset is requested up front.

Divergence note: with `schema=None` the column types come from Arrow C++'s CSV / JSON type inference, which is close to
but not the same code as arrow-rs' `Format::infer_schema` (csv.rs:58); pass an explicit schema for exact control."""
from __future__ import annotations

from typing import Optional, Sequence

import pyarrow as pa

from .plan import MemoryTable


class CsvReadOptions:
    """datasource/file/csv.rs:15-31. Defaults: header row, ',' delimiter; quote / escape None = arrow-rs' Format defaults
    (csv.rs:42-51 only overrides them when given): fields may be quoted with '"', a quote inside is doubled, no escape
    character."""

    def __init__(self, has_header: bool = True, delimiter: str = ",", quote: Optional[str] = None, escape: Optional[str] = None):
        self.has_header, self.delimiter, self.quote, self.escape = has_header, delimiter, quote, escape


def _table_of(tbl: pa.Table, batch_rows: Optional[int]) -> MemoryTable:
    tbl = tbl.combine_chunks()
    batches = tbl.to_batches(max_chunksize=batch_rows) if batch_rows else tbl.to_batches()
    return MemoryTable(tbl.schema, batches, lazy_upload=True)


def read_csv(path: str, options: Optional[CsvReadOptions] = None, schema: Optional[pa.Schema] = None,
             batch_rows: Optional[int] = 1 << 20) -> MemoryTable:
    """read_csv (datasource/file/csv.rs:33-70): the whole file becomes one in-memory table"""
    import pyarrow.csv as pacsv
    o = options or CsvReadOptions()
    names = [f.name for f in schema] if (schema is not None and not o.has_header) else None
    read = pacsv.ReadOptions(autogenerate_column_names=(not o.has_header and names is None), column_names=names)
    parse = pacsv.ParseOptions(delimiter=o.delimiter, quote_char=o.quote if o.quote else '"', escape_char=o.escape if o.escape else False)
    conv = pacsv.ConvertOptions(column_types=schema) if schema is not None else pacsv.ConvertOptions()
    return _table_of(pacsv.read_csv(path, read_options=read, parse_options=parse, convert_options=conv), batch_rows)


def read_parquet(path: str, columns: Optional[Sequence[str]] = None, batch_rows: Optional[int] = 1 << 20) -> MemoryTable:
    """read_parquet (datasource/file/parquet.rs): `columns` (optional) is a projection applied while reading the file"""
    import pyarrow.parquet as pq
    return _table_of(pq.read_table(path, columns=list(columns) if columns else None), batch_rows)


def read_json(path: str, batch_rows: Optional[int] = 1 << 20) -> MemoryTable:
    """read_json (datasource/file/json.rs): newline-delimited JSON"""
    import pyarrow.json as pajson
    return _table_of(pajson.read_json(path), batch_rows)
