"""qurious_amd — MI355X (gfx950) execution backend for the filter / hash-aggregate / hash-join path of
holicc/qurious. ``csrc/`` holds the hand-written HIP kernels and the C ABI (include/qhip.h); the Python
modules mirror the reference's operator interface for that path and call the C ABI through ctypes.

Import order matters in mixed processes: importing this package first makes libqhip.so bind the system
ROCm runtime (/opt/rocm/lib); a later ``import torch`` then shares that runtime (same sonames).
"""
from ._ffi import (ArrowError, Context, DeviceTable, HipError, InternalError, QuriousError, UnsupportedError,  # noqa: F401
                   get_context, load_library)
from .datatypes import JoinSide, JoinType, Operator, ScalarValue  # noqa: F401
from .expr import (AvgAggregateExpr, BinaryExpr, CaseExpr, CastExpr, Column, CountAggregateExpr, IsNotNull, IsNull,  # noqa: F401
                   Like, Literal, MaxAggregateExpr, MinAggregateExpr, Negative, PhysicalExpr, SumAggregateExpr, avg_return_type)
from .planner import DefaultQueryPlanner  # noqa: F401
from .datasource import CsvReadOptions, read_csv, read_json, read_parquet  # noqa: F401
from .plan import (CrossJoin, Filter, HashAggregate, HashJoinExec, JoinFilter, Limit, MemoryTable, NestedLoopJoinExec,  # noqa: F401
                   NoGroupingAggregate,
                   PhysicalPlan, PhysicalSortExpr, Projection, Scan, Sort, SortOptions, build_join_schema)

__version__ = "0.1.0"
