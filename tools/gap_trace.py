"""Where the time between two queries goes (QHIP_TRACE=2 prints absolute host times of a few points in libqhip): Q3 at SF10,
a few steady-state iterations. usage: QHIP_TRACE=2 python tools/gap_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth
c, o, l = synth.q3_tables(10.0)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
for _ in range(6):
    plan.execute_device()
print("=== steady state", file=sys.stderr, flush=True)
for _ in range(4):
    plan.execute_device()
