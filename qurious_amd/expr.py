"""PhysicalExpr mirrors (physical/expr/{column,literal,binary,cast,is_null,is_not_null,negative}.rs and
physical/expr/aggregate/*.rs). These classes only DESCRIBE expressions; evaluation happens inside the
HIP kernels that libqhip instantiates for the plan (there is no ``evaluate`` running on the CPU here)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import pyarrow as pa

from . import _ffi
from .datatypes import Operator, ScalarValue, to_qhip_dtype

# qhip_expr_kind
K_COLUMN, K_LITERAL, K_BINARY, K_CAST, K_IS_NULL, K_IS_NOT_NULL, K_NEGATIVE, K_IF, K_LIKE = range(9)
# qhip_agg_kind
AGG_SUM, AGG_AVG, AGG_COUNT, AGG_MIN, AGG_MAX = range(5)


class PhysicalExpr:
    """trait PhysicalExpr (physical/expr/mod.rs:33-35)."""

    def _lower(self, out: "ExprArray") -> int:
        raise NotImplementedError


class Column(PhysicalExpr):
    """physical/expr/column.rs:8-34"""

    def __init__(self, name: str, index: int):
        self.name, self.index = name, index

    def _lower(self, out):
        return out.add(kind=K_COLUMN, column=self.index)

    def __str__(self):
        return f"{self.name}({self.index})"


class Literal(PhysicalExpr):
    """physical/expr/literal.rs:8-22"""

    def __init__(self, value: ScalarValue):
        self.value = value

    def _lower(self, out):
        v = self.value
        dt = to_qhip_dtype(v.dtype)
        kw = dict(kind=K_LITERAL, dtype=dt, lit_is_null=1 if v.value is None else 0)
        if v.value is not None:
            t = v.dtype
            if pa.types.is_string(t):
                raw = v.value.encode() if isinstance(v.value, str) else bytes(v.value)
                kw.update(lit_str=out.keep(raw), lit_len=len(raw))
            elif pa.types.is_floating(t):
                kw.update(lit_f64=float(v.value))
            elif pa.types.is_boolean(t):
                kw.update(lit_lo=1 if v.value else 0)
            else:
                iv = int(v.value)
                kw.update(lit_lo=iv & 0xFFFFFFFFFFFFFFFF, lit_hi=_s64((iv >> 64) & 0xFFFFFFFFFFFFFFFF))
        return out.add(**kw)

    def __str__(self):
        return str(self.value)


def _s64(u: int) -> int:
    return u - (1 << 64) if u >= (1 << 63) else u


class BinaryExpr(PhysicalExpr):
    """physical/expr/binary.rs:17-70"""

    def __init__(self, left: PhysicalExpr, op: Operator, right: PhysicalExpr):
        self.left, self.op, self.right = left, Operator(op), right

    def _lower(self, out):
        l = self.left._lower(out)
        r = self.right._lower(out)
        return out.add(kind=K_BINARY, op=int(self.op), left=l, right=r)

    def __str__(self):
        return f"{self.left} {self.op} {self.right}"


class CastExpr(PhysicalExpr):
    """physical/expr/cast.rs:20-37 (CastOptions { safe: false })"""

    def __init__(self, expr: PhysicalExpr, data_type: pa.DataType):
        self.expr, self.data_type = expr, data_type

    def _lower(self, out):
        c = self.expr._lower(out)
        return out.add(kind=K_CAST, left=c, dtype=to_qhip_dtype(self.data_type))

    def __str__(self):
        return f"CAST({self.expr} AS {self.data_type})"


class IsNull(PhysicalExpr):
    def __init__(self, expr):
        self.expr = expr

    def _lower(self, out):
        return out.add(kind=K_IS_NULL, left=self.expr._lower(out))


class IsNotNull(PhysicalExpr):
    def __init__(self, expr):
        self.expr = expr

    def _lower(self, out):
        return out.add(kind=K_IS_NOT_NULL, left=self.expr._lower(out))


class Negative(PhysicalExpr):
    def __init__(self, expr):
        self.expr = expr

    def _lower(self, out):
        return out.add(kind=K_NEGATIVE, left=self.expr._lower(out))


class CaseExpr(PhysicalExpr):
    """physical/expr/case.rs:13-48: searched CASE as nested zip(mask, truthy, falsy), folded from the ELSE branch up"""

    def __init__(self, when_then: Sequence, else_expr: PhysicalExpr):
        self.when_then, self.else_expr = [(w, t) for w, t in when_then], else_expr

    def _lower(self, out):
        acc = self.else_expr._lower(out)
        for when, then in reversed(self.when_then):
            acc = out.add(kind=K_IF, left=when._lower(out), right=then._lower(out), third=acc)
        return acc

    def __str__(self):
        return "CASE" + "".join(f" WHEN {w} THEN {t}" for w, t in self.when_then) + f" ELSE {self.else_expr} END"


class Like(PhysicalExpr):
    """physical/expr/like.rs:14-43: arrow `like` / `nlike` (% any sequence, _ one character, backslash escapes)"""

    def __init__(self, negated: bool, expr: PhysicalExpr, pattern: PhysicalExpr):
        self.negated, self.expr, self.pattern = bool(negated), expr, pattern

    def _lower(self, out):
        return out.add(kind=K_LIKE, op=int(self.negated), left=self.expr._lower(out), right=self.pattern._lower(out))

    def __str__(self):
        return f"{self.expr} {'NOT LIKE' if self.negated else 'LIKE'} {self.pattern}"


class ExprArray:
    """Flat qhip_expr array under construction."""

    def __init__(self):
        self.nodes: List[_ffi.qhip_expr] = []
        self._keep = []

    def keep(self, raw: bytes):
        buf = C.create_string_buffer(raw, len(raw) + 1)
        self._keep.append(buf)
        return C.cast(buf, C.c_char_p)

    def add(self, **kw) -> int:
        e = _ffi.qhip_expr()
        e.kind, e.op, e.column, e.left, e.right = kw.get("kind", 0), kw.get("op", 0), kw.get("column", -1), kw.get("left", -1), kw.get("right", -1)
        e.third = kw.get("third", -1)
        if "dtype" in kw:
            e.dtype = kw["dtype"]
        e.lit_is_null = kw.get("lit_is_null", 0)
        e.lit_lo = kw.get("lit_lo", 0)
        e.lit_hi = kw.get("lit_hi", 0)
        e.lit_f64 = kw.get("lit_f64", 0.0)
        e.lit_str = kw.get("lit_str", None)
        e.lit_len = kw.get("lit_len", 0)
        self.nodes.append(e)
        return len(self.nodes) - 1

    def lower(self, expr: PhysicalExpr) -> int:
        return expr._lower(self)

    def c_array(self):
        n = max(1, len(self.nodes))
        arr = (_ffi.qhip_expr * n)()
        for k, e in enumerate(self.nodes):
            arr[k] = e
        return arr, len(self.nodes)


def int32_array(values: Sequence[int]):
    n = max(1, len(values))
    return (C.c_int32 * n)(*values)


# ---------------------------------------------------------------- aggregate expressions
class AggregateExpr:
    """trait AggregateExpr (physical/expr/aggregate/mod.rs:16-19)."""
    kind = -1

    def expression(self) -> PhysicalExpr:
        return self.expr

    def _return_type(self) -> pa.DataType:
        return self.return_type


class SumAggregateExpr(AggregateExpr):
    """aggregate/sum.rs:13-22"""
    kind = AGG_SUM

    def __init__(self, expr: PhysicalExpr, return_type: pa.DataType):
        self.expr, self.return_type = expr, return_type

    def __str__(self):
        return f"SUM({self.expr})"


class AvgAggregateExpr(AggregateExpr):
    """aggregate/avg.rs:15-29"""
    kind = AGG_AVG

    def __init__(self, expr: PhysicalExpr, expr_data_type: pa.DataType, return_type: pa.DataType):
        self.expr, self.expr_data_type, self.return_type = expr, expr_data_type, return_type


class CountAggregateExpr(AggregateExpr):
    """aggregate/count.rs:8-17"""
    kind = AGG_COUNT

    def __init__(self, expr: PhysicalExpr):
        self.expr, self.return_type = expr, pa.int64()

    def __str__(self):
        return f"COUNT({self.expr})"


class MinAggregateExpr(AggregateExpr):
    """aggregate/min.rs:30-40"""
    kind = AGG_MIN

    def __init__(self, expr: PhysicalExpr, return_type: pa.DataType):
        self.expr, self.return_type = expr, return_type


class MaxAggregateExpr(AggregateExpr):
    """aggregate/max.rs:29-39"""
    kind = AGG_MAX

    def __init__(self, expr: PhysicalExpr, return_type: pa.DataType):
        self.expr, self.return_type = expr, return_type


def avg_return_type(t: pa.DataType) -> pa.DataType:
    """logical/expr/aggregate.rs:75-90"""
    if pa.types.is_decimal128(t):
        return pa.decimal128(min(38, t.precision + 4), min(38, t.scale + 4))
    if pa.types.is_integer(t) or pa.types.is_floating(t):
        return pa.float64()
    raise _ffi.InternalError(_ffi.QHIP_INVALID_ARGUMENT, f"avg does not support {t}")
