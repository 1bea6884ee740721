"""ScalarValue / Operator / JoinType mirrors (datatypes/scalar.rs:85-107, datatypes/operator.rs:4-20,
common/join_type.rs:4-11) and the pyarrow <-> qhip_dtype mapping."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Any, Optional

import pyarrow as pa

from . import _ffi

# qhip_type_id
(T_NULL, T_BOOL, T_INT8, T_INT16, T_INT32, T_INT64, T_UINT8, T_UINT16, T_UINT32, T_UINT64, T_FLOAT32, T_FLOAT64,
 T_DATE32, T_DATE64, T_DECIMAL128, T_UTF8, T_TIME32_S, T_TIME32_MS, T_TIME64_US, T_TIME64_NS,
 T_TIMESTAMP_S, T_TIMESTAMP_MS, T_TIMESTAMP_US, T_TIMESTAMP_NS) = range(24)

_PA_TO_ID = [
    (pa.types.is_null, T_NULL), (pa.types.is_boolean, T_BOOL), (pa.types.is_int8, T_INT8), (pa.types.is_int16, T_INT16),
    (pa.types.is_int32, T_INT32), (pa.types.is_int64, T_INT64), (pa.types.is_uint8, T_UINT8), (pa.types.is_uint16, T_UINT16),
    (pa.types.is_uint32, T_UINT32), (pa.types.is_uint64, T_UINT64), (pa.types.is_float32, T_FLOAT32),
    (pa.types.is_float64, T_FLOAT64), (pa.types.is_date32, T_DATE32), (pa.types.is_date64, T_DATE64),
    (pa.types.is_decimal128, T_DECIMAL128), (pa.types.is_string, T_UTF8),
    (lambda t: pa.types.is_time32(t) and t.unit == "s", T_TIME32_S), (lambda t: pa.types.is_time32(t) and t.unit == "ms", T_TIME32_MS),
    (lambda t: pa.types.is_time64(t) and t.unit == "us", T_TIME64_US), (lambda t: pa.types.is_time64(t) and t.unit == "ns", T_TIME64_NS),
    # Timestamp(unit, None): MIN / MAX / comparison / sort-key type (aggregate/mod.rs:108-111); a timezone-qualified one is unsupported
    (lambda t: pa.types.is_timestamp(t) and t.tz is None and t.unit == "s", T_TIMESTAMP_S),
    (lambda t: pa.types.is_timestamp(t) and t.tz is None and t.unit == "ms", T_TIMESTAMP_MS),
    (lambda t: pa.types.is_timestamp(t) and t.tz is None and t.unit == "us", T_TIMESTAMP_US),
    (lambda t: pa.types.is_timestamp(t) and t.tz is None and t.unit == "ns", T_TIMESTAMP_NS),
]


def to_qhip_dtype(t: pa.DataType) -> _ffi.qhip_dtype:
    for pred, tid in _PA_TO_ID:
        if pred(t):
            if tid == T_DECIMAL128:
                return _ffi.qhip_dtype(tid, t.precision, t.scale)
            return _ffi.qhip_dtype(tid, 0, 0)
    raise _ffi.UnsupportedError(_ffi.QHIP_UNSUPPORTED, f"data type {t} is not supported by the HIP backend")


class Operator(enum.IntEnum):
    """datatypes/operator.rs:4-20 (same order as qhip_operator)."""
    Eq = 0
    NotEq = 1
    Gt = 2
    GtEq = 3
    Lt = 4
    LtEq = 5
    And = 6
    Or = 7
    Add = 8
    Sub = 9
    Mul = 10
    Div = 11
    Mod = 12

    def __str__(self):
        return ["=", "!=", ">", ">=", "<", "<=", "AND", "OR", "+", "-", "*", "/", "%"][int(self)]


class JoinType(enum.IntEnum):
    """common/join_type.rs:4-11 (same order as qhip_join_type)."""
    Left = 0
    Right = 1
    Inner = 2
    Full = 3
    LeftSemi = 4
    LeftAnti = 5


class JoinSide(enum.IntEnum):
    Left = 0
    Right = 1


@dataclass(frozen=True)
class ScalarValue:
    """datatypes/scalar.rs:85-107. ``value is None`` is the typed NULL of that variant."""
    dtype: pa.DataType
    value: Optional[Any] = None

    # constructors named like the Rust variants
    @staticmethod
    def Null():
        return ScalarValue(pa.null(), None)

    @staticmethod
    def Boolean(v=None):
        return ScalarValue(pa.bool_(), v)

    @staticmethod
    def Int8(v=None):
        return ScalarValue(pa.int8(), v)

    @staticmethod
    def Int16(v=None):
        return ScalarValue(pa.int16(), v)

    @staticmethod
    def Int32(v=None):
        return ScalarValue(pa.int32(), v)

    @staticmethod
    def Int64(v=None):
        return ScalarValue(pa.int64(), v)

    @staticmethod
    def UInt8(v=None):
        return ScalarValue(pa.uint8(), v)

    @staticmethod
    def UInt16(v=None):
        return ScalarValue(pa.uint16(), v)

    @staticmethod
    def UInt32(v=None):
        return ScalarValue(pa.uint32(), v)

    @staticmethod
    def UInt64(v=None):
        return ScalarValue(pa.uint64(), v)

    @staticmethod
    def Float32(v=None):
        return ScalarValue(pa.float32(), v)

    @staticmethod
    def Float64(v=None):
        return ScalarValue(pa.float64(), v)

    @staticmethod
    def Utf8(v=None):
        return ScalarValue(pa.string(), v)

    @staticmethod
    def Date32(v=None):
        """days since the epoch (not a Rust variant; produced by folding CAST(Utf8 AS Date32))"""
        return ScalarValue(pa.date32(), v)

    @staticmethod
    def Decimal128(v, precision: int, scale: int):
        """``v`` is the unscaled i128 (or None)."""
        return ScalarValue(pa.decimal128(precision, scale), v)

    def data_type(self) -> pa.DataType:
        return self.dtype

    def __str__(self):
        return "NULL" if self.value is None else f"{self.value}"
