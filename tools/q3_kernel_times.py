"""Device time of every kernel class of one Q3 query (HIP events are per operator; this uses QHIP stats per stage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth
from qurious_amd.exchange import DeviceSource

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ctx = q.get_context()
ctx.set_timing(True)
c, o, l = synth.q3_tables(sf)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
j2 = plan.input
j1 = j2.left
best = None
for it in range(6):
    a = j1.execute_device(); ctx.synchronize(); s1 = ctx.last_stats()
    j2b = q.HashJoinExec.try_new(DeviceSource(j1.schema(), a), j2.right, j2.join_type, j2.on, None)
    b = j2b.execute_device(); ctx.synchronize(); s2 = ctx.last_stats()
    cur = (s1["total_device_ms"], s1["main_kernel_ms"], s2["total_device_ms"], s2["main_kernel_ms"], s1["build_ms"], s2["build_ms"])
    best = cur if best is None else tuple(min(x, y) for x, y in zip(best, cur))
print(f"wgs={os.environ.get('QHIP_JOIN_SCATTER_WGS', '-')} J1 total {best[0]*1e3:.0f} us build {best[4]*1e3:.0f} probe {best[1]*1e3:.0f} us | J2 total {best[2]*1e3:.0f} us build {best[5]*1e3:.0f} probe {best[3]*1e3:.0f} us | rows {a.num_rows} {b.num_rows}", flush=True)
