"""ctypes binding of libqhip.so (include/qhip.h) — the only way Python reaches the HIP backend.

There is deliberately no fallback: if the shared library is missing this module raises at import
of the symbol table, and if no gfx950 device is visible ``get_context()`` raises. Nothing under
``oracle/`` is ever imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqhip.so")

# status codes (include/qhip.h)
QHIP_OK, QHIP_INVALID_ARGUMENT, QHIP_UNSUPPORTED, QHIP_HIP_ERROR, QHIP_OUT_OF_MEMORY, QHIP_EXEC_ERROR, QHIP_RCCL_ERROR, QHIP_RETRY = range(8)


class QuriousError(Exception):
    """Mirror of qurious::error::Error (error.rs:5-31); ``code`` is the qhip_status."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class InternalError(QuriousError):
    pass


class ArrowError(QuriousError):
    pass


class UnsupportedError(QuriousError):
    """QHIP_UNSUPPORTED: a valid plan the HIP backend does not accelerate (a Rust shim would fall back to the CPU node)."""


class HipError(QuriousError):
    pass


class RetryInput(QuriousError):
    """QHIP_RETRY: a hash join below ran without waiting for its output size and its room did not hold — the operator
    that reports it re-executes its input (plan.py: `_retrying`); never reaches a caller of ``execute()``."""


def _raise(code: int, msg: str):
    if code == QHIP_RETRY:
        raise RetryInput(code, msg)
    if code == QHIP_UNSUPPORTED:
        raise UnsupportedError(code, msg)
    if code in (QHIP_HIP_ERROR, QHIP_OUT_OF_MEMORY, QHIP_RCCL_ERROR):
        raise HipError(code, msg)
    if code == QHIP_EXEC_ERROR or msg.startswith("Arrow error") or msg.startswith("Invalid argument error"):
        raise ArrowError(code, msg)
    raise InternalError(code, msg)


class qhip_dtype(C.Structure):
    _fields_ = [("id", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32)]


class qhip_expr(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("op", C.c_int32), ("column", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
        ("third", C.c_int32), ("dtype", qhip_dtype), ("lit_is_null", C.c_int32),
        ("lit_lo", C.c_uint64), ("lit_hi", C.c_int64), ("lit_f64", C.c_double),
        ("lit_str", C.c_char_p), ("lit_len", C.c_int64),
    ]


class qhip_agg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("expr", C.c_int32), ("return_type", qhip_dtype)]


class qhip_exec_stats(C.Structure):
    _fields_ = [
        ("main_kernel_ms", C.c_double), ("total_device_ms", C.c_double), ("jit_ms", C.c_double),
        ("rows_in", C.c_int64), ("rows_out", C.c_int64), ("groups", C.c_int64), ("table_capacity", C.c_int64),
        ("retries", C.c_int32), ("lds_table_slots", C.c_int32), ("main_kernel_name", C.c_char * 64),
        ("lds_occupancy", C.c_double), ("hbm_table_load", C.c_double), ("lds_spilled", C.c_int32), ("workgroups", C.c_int32),
        ("bytes_per_row_read", C.c_double), ("build_ms", C.c_double), ("build_rows", C.c_int64), ("build_bytes_per_row", C.c_double),
    ]


class ArrowSchemaStruct(C.Structure):
    pass


class ArrowArrayStruct(C.Structure):
    pass


ArrowSchemaStruct._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchemaStruct))),
    ("dictionary", C.POINTER(ArrowSchemaStruct)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]
ArrowArrayStruct._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArrayStruct))),
    ("dictionary", C.POINTER(ArrowArrayStruct)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]

_lib = None
_lib_lock = threading.Lock()


def load_library() -> C.CDLL:
    """Load libqhip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C qurious_amd/csrc``)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP backend has no Python/CPU fallback)")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        P = C.POINTER
        sigs = {
            "qhip_ctx_create": (C.c_int, [C.c_int, P(vp)]),
            "qhip_ctx_destroy": (None, [vp]),
            "qhip_last_error": (C.c_char_p, [vp]),
            "qhip_version": (C.c_char_p, []),
            "qhip_device_available": (C.c_int, []),
            "qhip_ctx_last_stats": (C.c_int, [vp, P(qhip_exec_stats)]),
            "qhip_ctx_synchronize": (C.c_int, [vp]),
            "qhip_ctx_sync_count": (C.c_uint64, [vp]),
            "qhip_ctx_set_timing": (C.c_int, [vp, i32]),
            "qhip_ctx_allow_deferred_sizes": (C.c_int, [vp, i32]),
            "qhip_ctx_forget_plans": (C.c_int, [vp]),
            "qhip_ctx_device_name": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
            "qhip_table_from_arrow": (C.c_int, [vp, vp, P(vp), i64, P(vp)]),
            "qhip_table_from_arrow_lazy": (C.c_int, [vp, vp, P(vp), i64, P(vp)]),
            "qhip_table_to_arrow": (C.c_int, [vp, vp, i64, vp, vp]),
            "qhip_table_num_batches": (i64, [vp]),
            "qhip_table_batch_offsets": (C.c_int, [vp, P(i64), i64]),
            "qhip_table_num_rows": (i64, [vp]),
            "qhip_table_num_columns": (i64, [vp]),
            "qhip_table_column_bytes": (i64, [vp, i64]),
            "qhip_table_destroy": (None, [vp]),
            "qhip_filter_execute": (C.c_int, [vp, vp, P(qhip_expr), i32, i32, P(i32), i32, P(vp)]),
            "qhip_hash_aggregate_execute": (C.c_int, [vp, vp, P(qhip_expr), i32, i32, P(i32), i32, P(qhip_agg), i32,
                                                      P(C.c_char_p), P(vp)]),
            "qhip_hash_join_execute": (C.c_int, [vp, vp, vp, i32, P(qhip_expr), i32, P(qhip_expr), i32, P(i32), P(i32), i32,
                                                 P(qhip_expr), i32, i32, P(i32), P(i32), i32, i32, i32, P(vp)]),
            "qhip_nested_loop_join_execute": (C.c_int, [vp, vp, vp, i32, P(qhip_expr), i32, i32, P(i32), P(i32), i32, P(vp)]),
            "qhip_cross_join_execute": (C.c_int, [vp, vp, vp, P(vp)]),
            "qhip_projection_execute": (C.c_int, [vp, vp, P(qhip_expr), i32, P(i32), i32, P(C.c_char_p), P(vp)]),
            "qhip_sort_execute": (C.c_int, [vp, vp, P(qhip_expr), i32, P(i32), P(i32), P(i32), i32, i64, P(vp)]),
            "qhip_limit_execute": (C.c_int, [vp, vp, i64, i64, P(vp)]),
            "qhip_partition_by_key": (C.c_int, [vp, vp, P(qhip_expr), i32, P(i32), i32, i32, P(vp)]),
            "qhip_partition_filtered": (C.c_int, [vp, vp, P(qhip_expr), i32, P(i32), i32, i32, P(i32), i32, P(vp)]),
            "qhip_table_forget_statistics": (C.c_int, [vp]),
            "qhip_table_aux_bytes": (i64, [vp]),
            "qhip_table_concat": (C.c_int, [vp, P(vp), i32, P(vp)]),
            "qhip_table_column_buffer": (C.c_int, [vp, i64, i32, P(vp), P(i64)]),
            "qhip_table_wire_meta": (C.c_int, [vp, vp, P(i64), i32]),
            "qhip_table_keep_columns": (C.c_int, [vp, vp, P(i32), i32, P(vp)]),
            "qhip_table_pack": (C.c_int, [vp, vp, vp, i64]),
            "qhip_table_unpack_concat": (C.c_int, [vp, P(C.c_char_p), P(qhip_dtype), i32, P(i64), P(vp), i32, P(vp)]),
            "qhip_jit_compile_to_cache": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


BENCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libqhip_bench.so")
_BENCH_LIB = None


def load_bench_library():
    """libqhip_bench.so (include/qhip_bench.h): the synthetic TPC-H-shaped table generators and the streaming-read yardstick —
    benchmark support, kept out of the product library"""
    global _BENCH_LIB
    if _BENCH_LIB is None:
        if not os.path.exists(BENCH_LIB_PATH):
            raise ImportError(f"{BENCH_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(BENCH_LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        for name, (res, args) in {
            "qhip_synth_lineitem": (C.c_int, [i64, i64] + [vp] * 9),
            "qhip_synth_customer": (C.c_int, [i64, i64, vp, vp, vp]),
            "qhip_synth_orders": (C.c_int, [i64, i64, i64, vp, vp, vp, vp]),
            "qhip_synth_q3_lineitem_count": (i64, [i64, i64]),
            "qhip_synth_q3_lineitem": (C.c_int, [i64, i64, vp, vp, vp, vp]),
            "qhip_bench_stream_read": (C.c_int, [i32, i64, i32, C.POINTER(C.c_double)]),
        }.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _BENCH_LIB = lib
    return _BENCH_LIB


class Context:
    """Owner of one qhip_ctx (one HIP device, one stream). One per process and device is enough."""

    def __init__(self, device: int = -1):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.qhip_ctx_create(device, C.byref(h))
        if rc != QHIP_OK:
            _raise(rc, self.lib.qhip_last_error(None).decode())
        self.handle = h

    def check(self, rc: int):
        if rc != QHIP_OK:
            _raise(rc, self.lib.qhip_last_error(self.handle).decode())

    def last_stats(self) -> dict:
        st = qhip_exec_stats()
        self.check(self.lib.qhip_ctx_last_stats(self.handle, C.byref(st)))
        d = {f[0]: getattr(st, f[0]) for f in qhip_exec_stats._fields_}
        d["main_kernel_name"] = st.main_kernel_name.decode()
        return d

    def synchronize(self):
        self.check(self.lib.qhip_ctx_synchronize(self.handle))

    def set_timing(self, on: bool):
        """qhip_exec_stats timings (HIP events around an operator's phases) on / off; off by default, they cost stream time."""
        self.check(self.lib.qhip_ctx_set_timing(self.handle, 1 if on else 0))

    def sync_count(self) -> int:
        """Host waits on the device made through the library so far (the difference around a plan = its round trips)."""
        return int(self.lib.qhip_ctx_sync_count(self.handle))

    def forget_plans(self):
        """Drop everything learnt about executed plans (lowered plans, join sizes, group counts); compiled kernels stay."""
        self.check(self.lib.qhip_ctx_forget_plans(self.handle))

    def allow_deferred_sizes(self, delta: int):
        self._allow_depth = 0 if delta == 0 else max(0, getattr(self, "_allow_depth", 0) + int(delta))
        self.lib.qhip_ctx_allow_deferred_sizes(self.handle, int(delta))

    def no_deferred_sizes(self):
        """Context manager: no hash join below may leave its size on the device. The multi-rank operators run under it: a
        join of deferred size can answer QHIP_RETRY on ONE rank only, whose re-execution of the input would repeat
        collectives the other ranks do not take part in."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            depth = getattr(self, "_allow_depth", 0)
            if depth:
                self.lib.qhip_ctx_allow_deferred_sizes(self.handle, -depth)
            try:
                yield
            finally:
                if depth:
                    self.lib.qhip_ctx_allow_deferred_sizes(self.handle, depth)
        return _cm()

    def measure_stream_read(self, nbytes: int = 1 << 32, iters: int = 5) -> float:
        """Achieved GB/s of a plain streaming-read kernel over `nbytes` of HBM (the practical bandwidth ceiling): benchmark
        support, libqhip_bench.so (include/qhip_bench.h), not the product library."""
        out = C.c_double(0.0)
        self.synchronize()
        rc = load_bench_library().qhip_bench_stream_read(int(os.environ.get("QHIP_DEVICE", "0")), int(nbytes), int(iters), C.byref(out))
        if rc != 0:
            raise HipError(QHIP_HIP_ERROR, f"qhip_bench_stream_read failed ({rc})")
        return out.value

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self.check(self.lib.qhip_ctx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def close(self):
        if self.handle:
            self.lib.qhip_ctx_destroy(self.handle)
            self.handle = None


_ctx = None


def get_context() -> Context:
    """Process-wide context on the current HIP device (LOCAL_RANK under torch.distributed launchers)."""
    global _ctx
    if _ctx is None:
        dev = int(os.environ.get("QHIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ctx = Context(dev)
    return _ctx


class DeviceTable:
    """HBM-resident Vec<RecordBatch> (qhip_table)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle

    @staticmethod
    def from_batches(ctx: Context, schema: pa.Schema, batches, lazy: bool = False) -> "DeviceTable":
        """Upload now (default) or, with lazy=True, hand the batches over and let every column move to HBM the first time
        an operator or an export reads it (qhip_table_from_arrow_lazy)."""
        lib = ctx.lib
        c_schema = ArrowSchemaStruct()
        schema._export_to_c(C.addressof(c_schema))
        arrays = [ArrowArrayStruct() for _ in batches]
        try:
            for b, a in zip(batches, arrays):
                if b.schema != schema:
                    b = pa.RecordBatch.from_arrays(b.columns, schema=schema)
                b._export_to_c(C.addressof(a))
            ptrs = (C.c_void_p * max(1, len(arrays)))(*[C.addressof(a) for a in arrays])
            out = C.c_void_p()
            fn = lib.qhip_table_from_arrow_lazy if lazy else lib.qhip_table_from_arrow
            ctx.check(fn(ctx.handle, C.addressof(c_schema), ptrs, len(arrays), C.byref(out)))
            return DeviceTable(ctx, out)
        finally:
            _release_schema(c_schema)
            for a in arrays:
                _release_array(a)

    def forget_statistics(self):
        """drop the column statistics and narrow copies libqhip has collected on this table (qhip_table_forget_statistics)"""
        self.ctx.check(self.ctx.lib.qhip_table_forget_statistics(self.handle))

    @property
    def aux_bytes(self) -> int:
        return int(self.ctx.lib.qhip_table_aux_bytes(self.handle))

    @property
    def num_rows(self) -> int:
        return self.ctx.lib.qhip_table_num_rows(self.handle)

    @property
    def num_batches(self) -> int:
        return self.ctx.lib.qhip_table_num_batches(self.handle)

    @property
    def num_columns(self) -> int:
        return self.ctx.lib.qhip_table_num_columns(self.handle)

    def column_bytes(self, col: int) -> int:
        return self.ctx.lib.qhip_table_column_bytes(self.handle, col)

    def batch_offsets(self) -> List[int]:
        n = self.num_batches + 1
        out = (C.c_int64 * n)()
        self.ctx.check(self.ctx.lib.qhip_table_batch_offsets(self.handle, out, n))
        return list(out)

    def to_batches(self):
        nb = self.num_batches
        if nb > 8:
            # many batches: ONE download of every row, sliced on the host (zero-copy) at the batch boundaries
            a, s = ArrowArrayStruct(), ArrowSchemaStruct()
            self.ctx.check(self.ctx.lib.qhip_table_to_arrow(self.ctx.handle, self.handle, -1, C.addressof(a), C.addressof(s)))
            whole = pa.RecordBatch._import_from_c(C.addressof(a), C.addressof(s))
            off = self.batch_offsets()
            return [whole.slice(off[b], off[b + 1] - off[b]) for b in range(nb)]
        out = []
        for b in range(nb):
            a = ArrowArrayStruct()
            s = ArrowSchemaStruct()
            self.ctx.check(self.ctx.lib.qhip_table_to_arrow(self.ctx.handle, self.handle, b, C.addressof(a), C.addressof(s)))
            out.append(pa.RecordBatch._import_from_c(C.addressof(a), C.addressof(s)))
        return out

    def schema(self) -> pa.Schema:
        s = ArrowSchemaStruct()
        self.ctx.check(self.ctx.lib.qhip_table_to_arrow(self.ctx.handle, self.handle, 0, None, C.addressof(s)))
        return pa.Schema._import_from_c(C.addressof(s))

    def close(self):
        if self.handle:
            self.ctx.lib.qhip_table_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p)


def _release_schema(s: ArrowSchemaStruct):
    if s.release:
        _RELEASE_FN(s.release)(C.addressof(s))


def _release_array(a: ArrowArrayStruct):
    if a.release:
        _RELEASE_FN(a.release)(C.addressof(a))
