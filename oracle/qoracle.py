"""Python face of the CPU oracle (oracle/qoracle.c) — TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this module; the
product package ``qurious_amd`` never does. ``execute(plan)`` walks the same plan objects the HIP backend
executes (the plan classes are pure descriptions) but runs them with the C restatement of the reference's
operators: physical/plan/{scan,filter}.rs + datasource/memory.rs:69-98, aggregate/{hash,no_grouping}.rs,
join/{hash_join,mod}.rs, utils/{array,batch}.rs. pyarrow (Arrow C++) is used only for generic
take / concat / array construction, never for decimal arithmetic (it differs from arrow-rs there).

Pinning: the filter / aggregate / hash-join / sort / limit / nested-loop / cross-join restatements are pinned by the
reference's own unit-test and .slt vectors (tests/golden/reference_vectors.json, tests/test_reference_goldens.py,
tests/test_sort_limit.py, tests/test_nlj_cross.py). CASE (arrow `zip`), LIKE (arrow `like`) and Projection have NO vectors
in the reference (only TPC-H outputs whose inputs are unavailable offline): for those three the parity is UNPINNED by the
reference's tests; the restatement is cross-checked against Arrow C++ instead (tests/test_projection_exprs.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np
import pyarrow as pa

from qurious_amd import _ffi as _pffi          # POD struct layouts only (qhip_expr / qhip_agg / qhip_dtype)
from qurious_amd.datatypes import JoinSide, JoinType, to_qhip_dtype
from qurious_amd.expr import ExprArray
from qurious_amd import plan as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class qo_col(C.Structure):
    _fields_ = [("type", _pffi.qhip_dtype), ("n", C.c_int64), ("values", C.c_void_p), ("valid", C.c_void_p),
                ("offsets", C.c_void_p), ("data", C.c_void_p), ("owned", C.c_int)]


class qo_agg_result(C.Structure):
    _fields_ = [("n_groups", C.c_int64), ("first_row", C.POINTER(C.c_int64)), ("agg_cols", C.POINTER(qo_col)), ("n_aggs", C.c_int)]


class qo_hasher(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("k0", "k1", "length", "v0", "v2", "v1", "v3", "tail", "ntail")]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.qo_last_error.restype = C.c_char_p
        L.qo_hasher_finish.restype = C.c_uint64
        L.qo_filter_indices.restype = C.c_int64
        L.qo_filter_indices.argtypes = [C.POINTER(qo_col), C.c_void_p]
        L.qo_hasher_init.argtypes = [C.POINTER(qo_hasher)]
        L.qo_hasher_write.argtypes = [C.POINTER(qo_hasher), C.c_char_p, C.c_size_t]
        L.qo_hasher_finish.argtypes = [C.POINTER(qo_hasher)]
        L.qo_join_map_with_capacity.restype = C.c_void_p
        L.qo_join_map_with_capacity.argtypes = [C.c_int64]
        L.qo_join_map_free.argtypes = [C.c_void_p]
        L.qo_join_map_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.qo_join_map_is_distinct.argtypes = [C.c_void_p]
        L.qo_join_map_len.argtypes = [C.c_void_p]
        L.qo_join_map_len.restype = C.c_int64
        L.qo_join_map_get.argtypes = [C.c_void_p, C.c_uint64]
        L.qo_join_map_get.restype = C.c_int64
        L.qo_join_map_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.qo_join_map_next.restype = C.POINTER(C.c_uint64)
        L.qo_join_map_get_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint64))]
        L.qo_join_map_get_matches.restype = C.c_int64
        L.qo_probe_hash_table.argtypes = [C.c_void_p, C.POINTER(qo_col), C.POINTER(qo_col), C.c_int, C.c_int64,
                                          C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]
        L.qo_probe_hash_table.restype = C.c_int64
        L.qo_adjust_right_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int64))]
        L.qo_adjust_right_indices.restype = C.c_int64
        L.qo_free.argtypes = [C.c_void_p]
        L.qo_create_hashes.argtypes = [C.POINTER(qo_col), C.c_int, C.c_int64, C.c_void_p]
        L.qo_eval.argtypes = [C.POINTER(_pffi.qhip_expr), C.c_int, C.c_int, C.POINTER(qo_col), C.c_int, C.c_int64, C.POINTER(qo_col)]
        L.qo_hash_aggregate.argtypes = [C.POINTER(qo_col), C.c_int, C.POINTER(qo_col), C.POINTER(_pffi.qhip_agg), C.c_int, C.c_int64,
                                        C.c_void_p, C.c_int64, C.POINTER(qo_agg_result)]
        L.qo_col_free.argtypes = [C.POINTER(qo_col)]
        L.qo_agg_result_free.argtypes = [C.POINTER(qo_agg_result)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().qo_last_error().decode())


# ---------------------------------------------------------------- SipHash-1-3 / create_hashes
def siphash13(data: bytes) -> int:
    """std::hash::DefaultHasher::new() fed `data` in one write."""
    h = qo_hasher()
    L = lib()
    L.qo_hasher_init(C.byref(h))
    L.qo_hasher_write(C.byref(h), data, C.c_size_t(len(data)))
    return L.qo_hasher_finish(C.byref(h))


def siphash13_chunks(chunks: Sequence[bytes]) -> int:
    h = qo_hasher()
    L = lib()
    L.qo_hasher_init(C.byref(h))
    for c in chunks:
        L.qo_hasher_write(C.byref(h), c, C.c_size_t(len(c)))
    return L.qo_hasher_finish(C.byref(h))


_NP = {pa.int8(): np.int8, pa.int16(): np.int16, pa.int32(): np.int32, pa.int64(): np.int64, pa.uint8(): np.uint8,
       pa.uint16(): np.uint16, pa.uint32(): np.uint32, pa.uint64(): np.uint64, pa.float32(): np.float32,
       pa.float64(): np.float64, pa.date32(): np.int32, pa.date64(): np.int64, pa.time32("s"): np.int32, pa.time32("ms"): np.int32,
       pa.time64("us"): np.int64, pa.time64("ns"): np.int64, pa.timestamp("s"): np.int64, pa.timestamp("ms"): np.int64,
       pa.timestamp("us"): np.int64, pa.timestamp("ns"): np.int64}


class Col:
    """A qo_col plus the numpy arrays keeping its memory alive."""

    def __init__(self, arr: Optional[pa.Array] = None):
        self.c = qo_col()
        self.keep = []
        if arr is not None:
            self._from_arrow(arr)

    def _from_arrow(self, arr: pa.Array):
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
        t = arr.type
        n = len(arr)
        self.c.type = to_qhip_dtype(t)
        self.c.n = n
        bufs = arr.buffers()
        off = arr.offset
        if pa.types.is_null(t):
            valid = np.zeros(max(n, 1), dtype=np.uint8)
            self._set("valid", valid)
            return
        if arr.null_count > 0 and bufs[0] is not None:
            bits = np.unpackbits(np.frombuffer(bufs[0], dtype=np.uint8), bitorder="little")[off:off + n]
            self._set("valid", np.ascontiguousarray(bits, dtype=np.uint8))
        if pa.types.is_boolean(t):
            bits = np.unpackbits(np.frombuffer(bufs[1], dtype=np.uint8), bitorder="little")[off:off + n] if n else np.zeros(0, np.uint8)
            self._set("values", np.ascontiguousarray(bits, dtype=np.uint8))
        elif pa.types.is_string(t):
            offs = np.frombuffer(bufs[1], dtype=np.int32)[off:off + n + 1].copy() if bufs[1] is not None else np.zeros(1, np.int32)
            data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
            self._set("offsets", np.ascontiguousarray(offs))
            self._set("data", np.ascontiguousarray(data))
        elif pa.types.is_decimal128(t):
            raw = np.frombuffer(bufs[1], dtype=np.uint64).reshape(-1, 2)[off:off + n] if n else np.zeros((0, 2), np.uint64)
            self._set("values", np.ascontiguousarray(raw))
        else:
            raw = np.frombuffer(bufs[1], dtype=_NP[t])[off:off + n] if n else np.zeros(0, _NP[t])
            self._set("values", np.ascontiguousarray(raw))

    def _set(self, field, arr: np.ndarray):
        if arr.size == 0:
            arr = np.zeros(2, dtype=arr.dtype)
        self.keep.append(arr)
        setattr(self.c, field, arr.ctypes.data)


def col_to_arrow(c: qo_col, t: pa.DataType) -> pa.Array:
    """Copy a qo_col produced by the oracle into a pyarrow array of type t."""
    n = c.n
    valid = None
    if c.valid:
        valid = np.ctypeslib.as_array(C.cast(c.valid, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
    vbuf = None
    nulls = 0
    if valid is not None and n and (valid == 0).any():
        nulls = int((valid == 0).sum())
        vbuf = pa.py_buffer(np.packbits(valid, bitorder="little").tobytes())
    if pa.types.is_null(t):
        return pa.nulls(n)
    if pa.types.is_boolean(t):
        v = np.ctypeslib.as_array(C.cast(c.values, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n]
        return pa.Array.from_buffers(t, n, [vbuf, pa.py_buffer(np.packbits(v, bitorder="little").tobytes())], null_count=nulls)
    if pa.types.is_string(t):
        offs = np.ctypeslib.as_array(C.cast(c.offsets, C.POINTER(C.c_int32)), shape=(n + 1,)).copy()
        nb = int(offs[n])
        data = np.ctypeslib.as_array(C.cast(c.data, C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb].copy()
        return pa.Array.from_buffers(t, n, [vbuf, pa.py_buffer(offs.tobytes()), pa.py_buffer(data.tobytes())], null_count=nulls)
    width = 16 if pa.types.is_decimal128(t) else np.dtype(_NP[t]).itemsize
    raw = np.ctypeslib.as_array(C.cast(c.values, C.POINTER(C.c_uint8)), shape=(max(n * width, 1),))[:n * width].copy()
    return pa.Array.from_buffers(t, n, [vbuf, pa.py_buffer(raw.tobytes())], null_count=nulls)


def _col_array(cols: Sequence[Col]):
    arr = (qo_col * max(1, len(cols)))()
    for k, c in enumerate(cols):
        arr[k] = c.c
    return arr


def create_hashes(arrays: Sequence[pa.Array]) -> np.ndarray:
    """utils/array.rs:190-210"""
    cols = [Col(a) for a in arrays]
    n = len(arrays[0]) if arrays else 0
    out = np.zeros(max(n, 1), dtype=np.uint64)
    _check(lib().qo_create_hashes(_col_array(cols), len(cols), n, out.ctypes.data))
    return out[:n]


def _result_type(arr: pa.Array, c: qo_col) -> pa.DataType:
    tid = c.type.id
    from qurious_amd import datatypes as D
    table = {D.T_BOOL: pa.bool_(), D.T_INT8: pa.int8(), D.T_INT16: pa.int16(), D.T_INT32: pa.int32(), D.T_INT64: pa.int64(),
             D.T_UINT8: pa.uint8(), D.T_UINT16: pa.uint16(), D.T_UINT32: pa.uint32(), D.T_UINT64: pa.uint64(),
             D.T_FLOAT32: pa.float32(), D.T_FLOAT64: pa.float64(), D.T_DATE32: pa.date32(), D.T_DATE64: pa.date64(),
             D.T_UTF8: pa.string(), D.T_NULL: pa.null(), D.T_TIME32_S: pa.time32("s"), D.T_TIME32_MS: pa.time32("ms"),
             D.T_TIME64_US: pa.time64("us"), D.T_TIME64_NS: pa.time64("ns"), D.T_TIMESTAMP_S: pa.timestamp("s"),
             D.T_TIMESTAMP_MS: pa.timestamp("ms"), D.T_TIMESTAMP_US: pa.timestamp("us"), D.T_TIMESTAMP_NS: pa.timestamp("ns")}
    if tid == D.T_DECIMAL128:
        return pa.decimal128(c.type.precision, c.type.scale)
    return table[tid]


def evaluate(expr, batch: pa.RecordBatch) -> pa.Array:
    """PhysicalExpr::evaluate(&RecordBatch) -> ArrayRef"""
    ea = ExprArray()
    root = ea.lower(expr)
    arr, n = ea.c_array()
    cols = [Col(batch.column(k)) for k in range(batch.num_columns)]
    out = qo_col()
    _check(lib().qo_eval(arr, n, root, _col_array(cols), len(cols), batch.num_rows, C.byref(out)))
    try:
        return col_to_arrow(out, _result_type(None, out))
    finally:
        lib().qo_col_free(C.byref(out))


def _eval_col(expr, batch: pa.RecordBatch):
    """evaluate, returning (qo_col owner handle, pa.Array)"""
    a = evaluate(expr, batch)
    return a


def filter_batch(batch: pa.RecordBatch, predicate) -> pa.RecordBatch:
    """predicate.evaluate + filter_record_batch (physical/plan/filter.rs:33-34, datasource/memory.rs:90-93)"""
    mask = evaluate(predicate, batch)
    if not pa.types.is_boolean(mask.type):
        raise OracleError(1, "filter predicate is not Boolean (as_boolean() would panic, filter.rs:34)")
    m = Col(mask)
    sel = np.zeros(max(batch.num_rows, 1), dtype=np.int64)
    cnt = lib().qo_filter_indices(C.byref(m.c), sel.ctypes.data)
    if cnt < 0:
        raise OracleError(1, lib().qo_last_error().decode())
    return batch.take(pa.array(sel[:cnt], type=pa.int64()))


def _concat(schema: pa.Schema, batches: Sequence[pa.RecordBatch]) -> pa.RecordBatch:
    """arrow::compute::concat_batches"""
    if not batches:
        return pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in schema], schema=schema)
    tbl = pa.Table.from_batches([b.cast(schema) if b.schema != schema else b for b in batches]).combine_chunks()
    arrays = [tbl.column(k).chunk(0) if tbl.column(k).num_chunks else pa.array([], type=schema.field(k).type) for k in range(tbl.num_columns)]
    return pa.RecordBatch.from_arrays(arrays, schema=schema)


def _aggregate(batch: pa.RecordBatch, batch_offsets: List[int], group_exprs, aggregate_exprs, out_schema: Optional[pa.Schema]):
    keys = [evaluate(g, batch) for g in group_exprs]
    args = [evaluate(a.expression(), batch) for a in aggregate_exprs]
    kc = [Col(k) for k in keys]
    ac = [Col(a) for a in args]
    aggs = (_pffi.qhip_agg * max(1, len(aggregate_exprs)))()
    for k, a in enumerate(aggregate_exprs):
        aggs[k].kind = a.kind
        aggs[k].expr = 0
        aggs[k].return_type = to_qhip_dtype(a._return_type())
    offs = np.asarray(batch_offsets, dtype=np.int64)
    res = qo_agg_result()
    _check(lib().qo_hash_aggregate(_col_array(kc), len(kc), _col_array(ac), aggs, len(aggregate_exprs), batch.num_rows,
                                   offs.ctypes.data, len(batch_offsets) - 1, C.byref(res)))
    try:
        G = res.n_groups
        first = np.ctypeslib.as_array(res.first_row, shape=(max(G, 1),))[:G].copy() if G else np.zeros(0, np.int64)
        cols = [k.take(pa.array(first, type=pa.int64())) for k in keys]
        for a, ae in enumerate(aggregate_exprs):
            cols.append(col_to_arrow(res.agg_cols[a], ae._return_type()))
    finally:
        lib().qo_agg_result_free(C.byref(res))
    if out_schema is not None and len(out_schema) == len(cols):
        fields = [pa.field(f.name, c.type, True) for f, c in zip(out_schema, cols)]
        return pa.RecordBatch.from_arrays(cols, schema=pa.schema(fields))
    return pa.RecordBatch.from_arrays(cols, names=[f"c{k}" for k in range(len(cols))])


# ---------------------------------------------------------------- join
class JoinHashMap:
    """physical/plan/join/hash_join.rs:39-107"""

    def __init__(self, capacity: int):
        self.h = lib().qo_join_map_with_capacity(capacity)
        self.capacity = capacity

    def update(self, hashes: np.ndarray, rows: Sequence[int], delete_offset: int = 0):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.int64))
        if hashes.size == 0:
            hashes = np.zeros(1, np.uint64)
        self._keep = hashes
        lib().qo_join_map_update(self.h, hashes.ctypes.data, rows.ctypes.data if rows.size else None, rows.size, delete_offset)

    def is_distinct(self) -> bool:
        return bool(lib().qo_join_map_is_distinct(self.h))

    def map_len(self) -> int:
        return lib().qo_join_map_len(self.h)

    def get(self, h: int) -> int:
        return lib().qo_join_map_get(self.h, h)

    def next(self) -> List[int]:
        n = C.c_int64()
        p = lib().qo_join_map_next(self.h, C.byref(n))
        return [int(p[k]) for k in range(n.value)]

    def get_matches_indices(self, probe_hashes):
        ph = np.ascontiguousarray(np.asarray(probe_hashes, dtype=np.uint64))
        pi, mi = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)()
        cnt = lib().qo_join_map_get_matches(self.h, ph.ctypes.data if ph.size else None, ph.size, C.byref(pi), C.byref(mi))
        a = [int(pi[k]) for k in range(cnt)]
        b = [int(mi[k]) for k in range(cnt)]
        lib().qo_free(pi)
        lib().qo_free(mi)
        return a, b

    def __del__(self):
        try:
            lib().qo_join_map_free(self.h)
        except Exception:
            pass


def _take_nullable(col: pa.Array, idx: np.ndarray) -> pa.Array:
    """compute::take with NULL indices (encoded as -1) -> NULL rows"""
    mask = idx < 0
    ia = pa.array(np.where(mask, 0, idx), type=pa.int64(), mask=mask)
    return col.take(ia)


def build_batch_from_indices(schema: pa.Schema, column_indices, build_batch, probe_batch, build_idx: np.ndarray, probe_idx: np.ndarray):
    """utils/batch.rs:18-61 (build side = JoinSide::Left)"""
    cols = []
    for (idx, side) in column_indices:
        if side == JoinSide.Left:
            col = build_batch.column(idx)
            if len(col) == 0 or col.null_count == len(col):
                cols.append(pa.nulls(len(build_idx), type=col.type))
            else:
                cols.append(_take_nullable(col, build_idx))
        else:
            col = probe_batch.column(idx)
            if len(col) == 0 or col.null_count == len(col):
                cols.append(pa.nulls(len(probe_idx), type=col.type))
            else:
                cols.append(_take_nullable(col, probe_idx))
    fields = [pa.field(f.name, f.type, True) for f in schema]
    return pa.RecordBatch.from_arrays(cols, schema=pa.schema(fields, metadata=schema.metadata))


def _hash_join(node: "P.HashJoinExec") -> List[pa.RecordBatch]:
    L = lib()
    left_batches = execute(node.left)
    left_schema = node.left.schema()
    build = _concat(left_schema, left_batches)                                   # hash_join.rs:154
    bkeys = [evaluate(l, build) for l, _ in node.on]                              # :155-158
    hashes = create_hashes(bkeys)                                                 # :161-162
    jm = JoinHashMap(build.num_rows)
    jm.update(hashes, list(range(build.num_rows - 1, -1, -1)), 0)                 # :164 (.rev())
    visited = np.zeros(build.num_rows, dtype=bool)                                # :166-167
    bk = [Col(k) for k in bkeys]
    out = []
    jt = node.join_type
    for rb in execute(node.right):                                                # :363
        pkeys = [evaluate(r, rb) for _, r in node.on]
        pk = [Col(k) for k in pkeys]
        bi, pi = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
        cnt = L.qo_probe_hash_table(jm.h, _col_array(bk), _col_array(pk), len(bk), rb.num_rows, C.byref(bi), C.byref(pi))
        if cnt < 0:
            raise OracleError(1, L.qo_last_error().decode())
        b_idx = np.array([bi[k] for k in range(cnt)], dtype=np.int64)
        p_idx = np.array([pi[k] for k in range(cnt)], dtype=np.int64)
        L.qo_free(bi)
        L.qo_free(pi)
        if node.filter is not None and not (len(b_idx) == 0 and len(p_idx) == 0):  # join/mod.rs:125-154
            inter = build_batch_from_indices(node.filter.schema, node.filter.column_indices, build, rb, b_idx, p_idx)
            mask = evaluate(node.filter.expr, inter)
            keep = np.array([bool(v.as_py()) if v.is_valid else False for v in mask], dtype=bool) if len(mask) else np.zeros(0, bool)
            b_idx, p_idx = b_idx[keep], p_idx[keep]
        if jt in (JoinType.Right, JoinType.Full):                                   # join/mod.rs:156-207
            ob, op = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
            bu = np.ascontiguousarray(b_idx.astype(np.uint64))
            pu = np.ascontiguousarray(p_idx.astype(np.uint32))
            m = L.qo_adjust_right_indices(bu.ctypes.data if bu.size else None, pu.ctypes.data if pu.size else None, len(bu), rb.num_rows,
                                          C.byref(ob), C.byref(op))
            b_idx = np.array([ob[k] for k in range(m)], dtype=np.int64)
            p_idx = np.array([op[k] for k in range(m)], dtype=np.int64)
            L.qo_free(ob)
            L.qo_free(op)
        visited[b_idx[b_idx >= 0]] = True                                           # hash_join.rs:253-255
        if jt in (JoinType.LeftSemi, JoinType.LeftAnti):                            # :260-262
            continue
        ob_ = build_batch_from_indices(node.schema(), node.column_indices, build, rb, b_idx, p_idx)
        if ob_.num_rows > 0:                                                        # :369-371
            out.append(ob_)
    empty_right = _concat(node.right.schema(), [])
    if jt == JoinType.LeftSemi:                                                     # :374-377, :314-343
        idx = np.nonzero(visited)[0].astype(np.int64)
        out.append(build_batch_from_indices(node.schema(), node.column_indices, build, empty_right, idx, np.full(len(idx), -1, np.int64)))
        return out
    if jt in (JoinType.Left, JoinType.Full, JoinType.LeftAnti):                     # :277-312
        idx = np.nonzero(~visited)[0].astype(np.int64)
        out.append(build_batch_from_indices(node.schema(), node.column_indices, build, empty_right, idx, np.full(len(idx), -1, np.int64)))
    return out


# ---------------------------------------------------------------- joins without equi-keys
def _nested_loop_join(node: "P.NestedLoopJoinExec") -> List[pa.RecordBatch]:
    """NestedLoopJoinExec::execute (nest_loop_join.rs:79-228), control flow and index order as in the reference"""
    jt = node.join_type
    lb = _concat(node.left.schema(), execute(node.left))                            # :83-84
    rb = _concat(node.right.schema(), execute(node.right))
    nl, nr = lb.num_rows, rb.num_rows
    build = lambda li, ri: build_batch_from_indices(node.schema(), node.column_indices, lb, rb, np.asarray(li, np.int64), np.asarray(ri, np.int64))  # noqa: E731
    if nr == 0:                                                                     # :87-121
        if jt in (JoinType.Inner, JoinType.Right):
            return []
        if jt == JoinType.LeftSemi:
            return [build([], [])]
        return [build(np.arange(nl), np.full(nl, -1))]                              # Left / Full / LeftAnti: every left row, NULL right
    # build_join_indices (:232-260): for every right row, all left rows, filtered by the JoinFilter
    li = np.tile(np.arange(nl, dtype=np.int64), nr)
    ri = np.repeat(np.arange(nr, dtype=np.int64), nl)
    if node.filter is not None and len(li):                                         # join_filter_indices (:262-289)
        f = node.filter
        inter = build_batch_from_indices(f.schema, f.column_indices, lb, rb, li, ri)
        mask = evaluate(f.expr, inter)
        keep = np.array([bool(v) for v in mask.fill_null(False).to_pylist()], dtype=bool)
        li, ri = li[keep], ri[keep]
    if jt in (JoinType.LeftSemi, JoinType.LeftAnti):                                # :141-170
        visited = np.zeros(nl, dtype=bool)
        visited[li] = True
        keep = np.nonzero(visited if jt == JoinType.LeftSemi else ~visited)[0]
        return [build(keep, np.full(len(keep), -1))]
    matched = build(li, ri)
    if jt == JoinType.Inner:                                                        # :172-174
        return [matched]
    l_un, r_un = np.zeros(0, np.int64), np.zeros(0, np.int64)
    if jt in (JoinType.Left, JoinType.Full):                                        # :193-204
        v = np.zeros(nl, dtype=bool)
        v[li] = True
        l_un = np.nonzero(~v)[0].astype(np.int64)
    if jt in (JoinType.Right, JoinType.Full):                                       # :206-217
        v = np.zeros(nr, dtype=bool)
        v[ri] = True
        r_un = np.nonzero(~v)[0].astype(np.int64)
    unmatched = build(np.concatenate([l_un, np.full(len(r_un), -1, np.int64)]), np.concatenate([np.full(len(l_un), -1, np.int64), r_un]))
    return [matched, unmatched]                                                     # :219-228


def _cross_join(node: "P.CrossJoin") -> List[pa.RecordBatch]:
    """CrossJoin::execute (cross_join.rs:121-166): per left batch, per right batch, per left row one output batch"""
    out = []
    right = execute(node.right)
    for lb in execute(node.left):
        for rb in right:
            for row in range(lb.num_rows):
                cols = [lb.column(k).take(pa.array(np.full(rb.num_rows, row, dtype=np.int64))) for k in range(lb.num_columns)] + list(rb.columns)
                out.append(pa.RecordBatch.from_arrays(cols, schema=pa.schema([pa.field(f.name, c.type, True) for f, c in zip(node.schema(), cols)])))
    return out


# ---------------------------------------------------------------- sort / limit
def _dense_rank(arr: pa.Array) -> np.ndarray:
    """Rank of every non-null value in arrow-ord's sort order (arrow 53 `sort`/`lexsort_to_indices`: integers, dates and
    decimals by value, floats by IEEE total order (`total_cmp`), Utf8 bytewise, false < true); NULL slots get 0."""
    t = arr.type
    n = len(arr)
    valid = np.asarray(arr.is_valid().to_numpy(zero_copy_only=False), dtype=bool) if arr.null_count else np.ones(n, dtype=bool)   # (no to_pylist: a timestamp may lie outside datetime's range)
    if pa.types.is_floating(t):
        bits = np.asarray(arr.cast(pa.float64()).fill_null(0.0).to_numpy(zero_copy_only=False), dtype=np.float64).view(np.int64)
        keys = np.where(bits < 0, ~bits, bits | np.int64(-2**63)).view(np.uint64).astype(object)     # total order image
    elif pa.types.is_string(t) or pa.types.is_large_string(t):
        keys = np.array([(v.encode() if v is not None else b"") for v in arr.to_pylist()], dtype=object)
    elif pa.types.is_decimal(t):
        sc = t.scale
        keys = np.array([(int(v.scaleb(sc)) if v is not None else 0) for v in arr.to_pylist()], dtype=object)
    elif pa.types.is_boolean(t):
        keys = np.array([(int(v) if v is not None else 0) for v in arr.to_pylist()], dtype=object)
    elif pa.types.is_date(t) or pa.types.is_time(t) or pa.types.is_timestamp(t):
        keys = np.array([0 if v is None else v for v in arr.cast(pa.int32() if (pa.types.is_date32(t) or pa.types.is_time32(t)) else pa.int64()).to_pylist()], dtype=object)
    elif pa.types.is_null(t):
        keys = np.zeros(n, dtype=object)
    else:
        keys = np.array([(v if v is not None else 0) for v in arr.to_pylist()], dtype=object)
    order = sorted(set(keys[valid].tolist()))
    rank_of = {k: r for r, k in enumerate(order)}
    return np.array([rank_of[k] if ok else 0 for k, ok in zip(keys.tolist(), valid.tolist())], dtype=np.int64), valid


def lexsort_to_indices(columns, limit: Optional[int] = None) -> np.ndarray:
    """arrow_ord::sort::lexsort_to_indices restated: columns = [(array, descending, nulls_first)], most significant
    first. `descending` reverses the values only; NULL placement follows nulls_first alone; equal rows keep no
    particular order in arrow — the reference adds the row number as the last key (sort.rs:62-73), which the caller
    passes like any other column."""
    n = len(columns[0][0]) if columns else 0
    keys = []
    for arr, descending, nulls_first in columns:
        rank, valid = _dense_rank(arr)
        if descending:
            rank = -rank
        null_key = np.where(valid, 1, 0) if nulls_first else np.where(valid, 0, 1)
        keys.append((null_key, np.where(valid, rank, 0)))
    flat = []
    for null_key, rank in reversed(keys):      # np.lexsort: last key is the primary one
        flat.append(rank)
        flat.append(null_key)
    idx = np.lexsort(flat) if flat else np.arange(n)
    return idx[:limit] if limit is not None else idx


def _sort(node: "P.Sort") -> pa.RecordBatch:
    schema = node.input.schema()
    merged = _concat(schema, execute(node.input))                                   # sort.rs:49
    cols = [(evaluate(e.expr, merged), e.options.descending, e.options.nulls_first) for e in node.exprs]   # sort.rs:50-60
    cols.append((pa.array(np.arange(merged.num_rows, dtype=np.uint64)), False, False))   # sort.rs:62-73: stable tie-break
    idx = lexsort_to_indices(cols, node.limit)                                      # sort.rs:75
    return merged.take(pa.array(idx.astype(np.uint64)))                             # sort.rs:76-80


def _limit(batches: List[pa.RecordBatch], fetch: Optional[int], skip: int) -> List[pa.RecordBatch]:
    """limit.rs:27-58, including the empty slice it emits when the window closes exactly on a batch boundary — and its
    OFFSET quirk: `skip` is only ever decremented by whole skipped batches (limit.rs:39-42), never cleared after the batch
    it was applied to (limit.rs:44), so with skip > 0 over several batches every LATER batch loses its first `skip` rows too
    (or is dropped whole, shrinking `skip`, when it has no more rows than that). Restated as written, not as SQL means it."""
    max_fetch = fetch if fetch is not None else (1 << 64) - 1
    results, fetched = [], 0
    for batch in batches:
        rows = batch.num_rows
        if rows <= skip:
            skip -= rows
            continue
        new_batch = batch.slice(skip, rows - skip)
        remaining = max_fetch - fetched
        if new_batch.num_rows <= remaining:
            results.append(new_batch)
            fetched += new_batch.num_rows
        else:
            results.append(new_batch.slice(0, remaining))
            break
    return results


# ---------------------------------------------------------------- plan walker
def execute(node) -> List[pa.RecordBatch]:
    """PhysicalPlan::execute() of the reference, on the CPU."""
    if isinstance(node, P.Scan):
        out = []
        for b in node.datasource.data:                                              # memory.rs:78-97
            if node.projections is not None:
                b = b.select(node.projections)
            if node.filter is not None:
                b = filter_batch(b, node.filter)
            out.append(b)
        return out
    if isinstance(node, P.HashJoinExec):
        return _hash_join(node)
    if isinstance(node, P.NestedLoopJoinExec):
        return _nested_loop_join(node)
    if isinstance(node, P.CrossJoin):
        return _cross_join(node)
    if isinstance(node, P.NoGroupingAggregate):
        batches = execute(node.input)                                               # no_grouping.rs:31
        offs = [0]
        for b in batches:
            offs.append(offs[-1] + b.num_rows)
        batch = _concat(node.input.schema(), batches)
        return [_aggregate(batch, offs, [], node.aggregate_exprs, node.schema())]
    if isinstance(node, P.HashAggregate):
        batches = execute(node.input)                                               # hash.rs:139
        if not batches:
            return []                                                               # hash.rs:146-148
        batch = _concat(batches[0].schema, batches)                                 # hash.rs:150
        return [_aggregate(batch, [0, batch.num_rows], node.group_exprs, node.aggregate_exprs, node.schema())]
    if isinstance(node, P.Filter):
        return [filter_batch(b, node.predicate) for b in execute(node.input)]       # filter.rs:29-43
    if isinstance(node, P.Projection):                                              # projection.rs:27-46
        out = []
        for b in execute(node.input):
            cols = [evaluate(e, b) for e in node.exprs]
            schema = node.schema() if node.schema() is not None else pa.schema([pa.field(f"c{k}", c.type) for k, c in enumerate(cols)])
            out.append(pa.RecordBatch.from_arrays(cols, schema=pa.schema([pa.field(f.name, c.type, True) for f, c in zip(schema, cols)])))
        return out
    if isinstance(node, P.Sort):
        return [_sort(node)]                                                        # sort.rs:81: always one batch
    if isinstance(node, P.Limit):
        return _limit(execute(node.input), node.fetch, node.skip)
    raise OracleError(2, f"oracle: unsupported plan node {type(node).__name__}")


# ---------------------------------------------------------------- timed whole-pipeline baseline (bench.py cpu_baseline)
def scan_filter_aggregate_timed(plan: "P.HashAggregate"):
    """Run HashAggregate(Scan(filter)) entirely inside the C restatement (qo_scan_filter_aggregate) and return
    (result RecordBatch, seconds of the C call, rows scanned). Single-threaded like the reference executor."""
    import time
    L = lib()
    L.qo_scan_filter_aggregate.argtypes = [C.POINTER(qo_col), C.c_int64, C.c_int, C.c_void_p, C.POINTER(_pffi.qhip_expr), C.c_int, C.c_int,
                                           C.c_void_p, C.c_int, C.POINTER(_pffi.qhip_agg), C.c_int, C.POINTER(qo_agg_result),
                                           C.POINTER(qo_col), C.POINTER(C.c_int64)]
    scan = plan.input
    assert isinstance(scan, P.Scan) and scan.projections is None
    data = scan.datasource.data
    ncols = len(scan.datasource.schema())
    cols = [Col(b.column(c)) for b in data for c in range(ncols)]
    carr = _col_array(cols)
    rows = np.asarray([b.num_rows for b in data], dtype=np.int64)
    ea = ExprArray()
    pred = ea.lower(scan.filter) if scan.filter is not None else -1
    groups = np.asarray([ea.lower(g) for g in plan.group_exprs], dtype=np.int32)
    aggs = (_pffi.qhip_agg * max(1, len(plan.aggregate_exprs)))()
    for k, a in enumerate(plan.aggregate_exprs):
        aggs[k].kind = a.kind
        aggs[k].expr = ea.lower(a.expression())
        aggs[k].return_type = to_qhip_dtype(a._return_type())
    arr, n = ea.c_array()
    res = qo_agg_result()
    okeys = (qo_col * max(1, len(groups)))()
    kept = C.c_int64(0)
    t0 = time.perf_counter()
    rc = L.qo_scan_filter_aggregate(carr, len(data), ncols, rows.ctypes.data if rows.size else None, arr, n, pred,
                                    groups.ctypes.data if groups.size else None, len(groups), aggs, len(plan.aggregate_exprs),
                                    C.byref(res), okeys, C.byref(kept))
    dt = time.perf_counter() - t0
    _check(rc)
    try:
        out_cols = [col_to_arrow(okeys[k], _result_type(None, okeys[k])) for k in range(len(groups))]
        for a, ae in enumerate(plan.aggregate_exprs):
            out_cols.append(col_to_arrow(res.agg_cols[a], ae._return_type()))
    finally:
        for k in range(len(groups)):
            L.qo_col_free(C.byref(okeys[k]))
        L.qo_agg_result_free(C.byref(res))
    batch = pa.RecordBatch.from_arrays(out_cols, names=[f"c{k}" for k in range(len(out_cols))])
    return batch, dt, int(rows.sum())


# ---------------------------------------------------------------- exchange mirror (tests of the multi-GPU path on CPUs)
def _mix64(x: np.ndarray) -> np.ndarray:
    x = x.copy()
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x


def partition_ids_by_range(key_array, upper_bounds: Sequence[int]) -> np.ndarray:
    """numpy restatement of qhip_partition_filtered_by_range's row -> part mapping (csrc/device/qhip_device.hpp qh_part_range):
    part = the number of bounds the key is GREATER than (bounds ascending; part p takes (bounds[p - 1], bounds[p]], the last part
    everything above); a NULL key goes where 0 goes."""
    a = key_array.combine_chunks() if isinstance(key_array, pa.ChunkedArray) else key_array
    t = a.type
    v = np.array(a.cast(pa.int32() if (pa.types.is_date32(t) or pa.types.is_time32(t)) else pa.int64()).fill_null(0)).astype(np.int64)
    return np.searchsorted(np.asarray(list(upper_bounds), dtype=np.int64), v, side="left").astype(np.int64)


def partition_ids(key_arrays: Sequence[pa.Array], n_parts: int) -> np.ndarray:
    """numpy restatement of qhip_partition_by_key's row -> part mapping (csrc/device/qhip_device.hpp qh_part_hash over the key
    words of csrc/codegen.cpp emit_key_words): ints/dates sign-extended to 64 bits, Decimal128 as (lo, hi), Utf8 <= 31 bytes
    packed into 4 words; a row with any NULL key has all its key words zeroed."""
    n = len(key_arrays[0])
    words = []
    valid = np.ones(n, dtype=bool)
    for a in key_arrays:
        if isinstance(a, pa.ChunkedArray):
            a = a.combine_chunks()
        if a.null_count:
            valid &= np.array(a.is_valid())
        t = a.type
        if pa.types.is_string(t):
            # the exchange always packs Utf8 keys into 4 words (<= 31 bytes): bytes little-endian, length in the top byte
            ws = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
            for i, s in enumerate(a.to_pylist()):
                if s is None:
                    continue
                b = s.encode()
                assert len(b) <= 31
                packed = int.from_bytes(b, "little") | (len(b) << 248)
                for k in range(4):
                    ws[k][i] = (packed >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
            words.extend(ws)
        elif pa.types.is_decimal128(t):
            raw = np.frombuffer(a.buffers()[1], dtype=np.uint64).reshape(-1, 2)[a.offset:a.offset + n]
            words.append(raw[:, 0].copy())
            words.append(raw[:, 1].copy())
        else:
            v = np.array(a.cast(pa.int32() if (pa.types.is_date32(t) or pa.types.is_time32(t)) else pa.int64()).fill_null(0)).astype(np.int64)
            words.append(v.view(np.uint64).copy())
    def fold_mul(x):   # qh_fold_mul: one 64-bit multiply, the high half folded into the low one
        x = x * np.uint64(0x9E3779B97F4A7C15)
        return x ^ (x >> np.uint64(32))
    with np.errstate(over="ignore"):
        z = [np.where(valid, w, np.uint64(0)) for w in words]
        h = fold_mul(z[0])
        for k, w in enumerate(z[1:], start=1):   # qh_key_hash
            h = fold_mul(h ^ w) + np.uint64(k)
        pid = ((h >> np.uint64(32)) * np.uint64(n_parts)) >> np.uint64(32)
    return pid.astype(np.int64)
