"""Plan-only helpers (no GPU needed): the kernel source libqhip instantiates for a plan, and ahead-of-time
hiprtc compilation of it into the on-disk kernel cache (include/qhip.h "plan-only entry points")."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import pyarrow as pa

from . import _ffi
from .datatypes import to_qhip_dtype
from .expr import AggregateExpr, ExprArray, PhysicalExpr, int32_array

KERNEL_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_kcache")


def _cols(schema: pa.Schema, has_nulls: Optional[Sequence[bool]]):
    n = len(schema)
    types = (_ffi.qhip_dtype * max(1, n))(*[to_qhip_dtype(f.type) for f in schema])
    hn = (C.c_int32 * max(1, n))(*[1 if (has_nulls and has_nulls[k]) else 0 for k in range(n)])
    return types, hn, n


def _call(fn, *args) -> str:
    lib = _ffi.load_library()
    lib.qhip_plan_last_error.restype = C.c_char_p
    need = C.c_size_t(0)
    rc = fn(*args, None, 0, C.byref(need))
    if rc != 0:
        _ffi._raise(rc, lib.qhip_plan_last_error().decode())
    buf = C.create_string_buffer(need.value)
    rc = fn(*args, buf, need.value, C.byref(need))
    if rc != 0:
        _ffi._raise(rc, lib.qhip_plan_last_error().decode())
    return buf.value.decode()


def aggregate_source(schema: pa.Schema, predicate: Optional[PhysicalExpr], group_exprs: Sequence[PhysicalExpr],
                     aggregate_exprs: Sequence[AggregateExpr], has_nulls: Optional[Sequence[bool]] = None) -> str:
    lib = _ffi.load_library()
    fn = lib.qhip_plan_aggregate_source
    fn.restype = C.c_int
    fn.argtypes = None
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    pred = ea.lower(predicate) if predicate is not None else -1
    groups = [ea.lower(g) for g in group_exprs]
    aggs = (_ffi.qhip_agg * max(1, len(aggregate_exprs)))()
    for k, a in enumerate(aggregate_exprs):
        aggs[k].kind = a.kind
        aggs[k].expr = ea.lower(a.expression())
        aggs[k].return_type = to_qhip_dtype(a._return_type())
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), C.c_int32(pred), int32_array(groups), C.c_int32(len(groups)), aggs,
                 C.c_int32(len(aggregate_exprs)))


def filter_source(schema: pa.Schema, predicate: PhysicalExpr, has_nulls: Optional[Sequence[bool]] = None) -> str:
    lib = _ffi.load_library()
    fn = lib.qhip_plan_filter_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    root = ea.lower(predicate)
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), C.c_int32(root))


def keys_source(schema: pa.Schema, keys: Sequence[PhysicalExpr], has_nulls: Optional[Sequence[bool]] = None) -> str:
    lib = _ffi.load_library()
    fn = lib.qhip_plan_keys_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)))


def probe_source(schema: pa.Schema, keys: Sequence[PhysicalExpr], predicate: Optional[PhysicalExpr] = None,
                 has_nulls: Optional[Sequence[bool]] = None) -> str:
    """Source of the fused probe kernel HashJoinExec launches for its probe (right) side."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_probe_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    proot = ea.lower(predicate) if predicate is not None else -1
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)), C.c_int32(proot))


def scatter_source(schema: pa.Schema, keys: Sequence[PhysicalExpr], predicate: Optional[PhysicalExpr] = None,
                   has_nulls: Optional[Sequence[bool]] = None) -> str:
    """Source of the build-side kernel of HashJoinExec's LDS-staged build (key words -> region entries)."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_scatter_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    proot = ea.lower(predicate) if predicate is not None else -1
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)), C.c_int32(proot))


def partition_source(schema: pa.Schema, keys: Sequence[PhysicalExpr], predicate: Optional[PhysicalExpr], n_parts: int,
                     has_nulls: Optional[Sequence[bool]] = None) -> str:
    """Source of pass 1 of the exchange's fused filter + partition (qhip_partition_filtered) for `n_parts` parts."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_partition_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    proot = ea.lower(predicate) if predicate is not None else -1
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)), C.c_int32(proot), C.c_int32(n_parts))


def part_scatter_source(widths: Sequence[int], n_parts: int, indirect: Optional[Sequence[bool]] = None) -> str:
    """Source of pass 2 of the exchange's fused filter + partition for one group of columns (bytes per value; 0 = row number)."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_part_scatter_source
    fn.restype = C.c_int
    n = len(widths)
    return _call(fn, int32_array(list(widths)), int32_array([1 if (indirect and indirect[k]) else 0 for k in range(n)]), C.c_int32(n), C.c_int32(n_parts))


def sort_keys_source(schema: pa.Schema, keys: Sequence[PhysicalExpr], has_nulls: Optional[Sequence[bool]] = None) -> str:
    """Source of the order-preserving key image kernel Sort launches (Utf8 keys have no generated part)."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_sort_keys_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)))


def projection_source(schema: pa.Schema, exprs: Sequence[PhysicalExpr], has_nulls: Optional[Sequence[bool]] = None) -> str:
    """Source of the kernel Projection launches for its computed (non-Column) expressions."""
    lib = _ffi.load_library()
    fn = lib.qhip_plan_projection_source
    fn.restype = C.c_int
    types, hn, n = _cols(schema, has_nulls)
    ea = ExprArray()
    roots = [ea.lower(e) for e in exprs]
    arr, ne = ea.c_array()
    return _call(fn, types, hn, C.c_int32(n), arr, C.c_int32(ne), int32_array(roots), C.c_int32(len(roots)))


def compile_to_cache(policy_source: str, cache_dir: str = KERNEL_CACHE_DIR) -> str:
    """hiprtc-compile (device templates + policy) for gfx950 into the kernel cache; returns the compiler log."""
    lib = _ffi.load_library()
    log = C.create_string_buffer(1 << 16)
    rc = lib.qhip_jit_compile_to_cache(policy_source.encode(), cache_dir.encode() if cache_dir else None, log, len(log))
    if rc != 0:
        _ffi._raise(rc, log.value.decode())
    return log.value.decode()
