"""Kernel catalog: the plan shapes of the benchmark configurations, compiled for gfx950 ahead of time.

``compile_catalog`` needs no GPU (hiprtc cross-compiles); it fills qurious_amd/_kcache so that the first
execution on a GPU box loads code objects instead of compiling. Plans outside the catalog are compiled by
libqhip on first use and cached the same way.
"""
from __future__ import annotations

import time

from . import planning, queries
from .plan import MemoryTable
from .synth import LINEITEM_SCHEMA


def catalog_sources():
    table = MemoryTable.try_new(LINEITEM_SCHEMA, [])
    out = []
    for name, plan in (("q1_mini", queries.q1_mini(table)), ("q1_full", queries.q1_full(table))):
        scan = plan.input
        out.append((name, planning.aggregate_source(LINEITEM_SCHEMA, scan.filter, plan.group_exprs, plan.aggregate_exprs)))
    return out


def compile_catalog(verbose: bool = False):
    for name, src in catalog_sources():
        t = time.time()
        planning.compile_to_cache(src)
        if verbose:
            print(f"[catalog] {name}: compiled for gfx950 in {time.time() - t:.2f}s")
