import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth
ctx = q.get_context()
table = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, synth.lineitem(100_000_000, 1 << 20))
plan = queries.q1_mini(table)
for _ in range(3): plan.execute_device()
ctx.synchronize()
os.environ["QHIP_TRACE"] = "1"
for _ in range(2):
    t = time.perf_counter(); out = plan.execute_device(); ctx.synchronize(); print("python-level step us", (time.perf_counter() - t) * 1e6, flush=True)
