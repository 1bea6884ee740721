"""Host-side cost of a plan: wall time of every C-ABI operator call of TPC-H Q3 (the joins of deferred size return as soon
as their work is enqueued, so their call time IS the host's enqueue cost; the aggregate's call contains the plan's one wait),
or of Q1's aggregate list over `rows` lineitem rows (a tiny table shows the fixed cost of a call).
usage: python tools/host_overhead.py [sf] | python tools/host_overhead.py q1 [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q  # noqa: E402
from qurious_amd import plan as P, queries, synth  # noqa: E402

ctx = q.get_context()
if len(sys.argv) > 1 and sys.argv[1] == "q1":
    rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 59_986_052
    sf = f"Q1 over {rows} rows"
    plan = queries.q1_full(q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, synth.lineitem(rows, 1 << 20)))
else:
    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    c, o, l = synth.q3_tables(sf)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    sf = f"Q3 SF{sf}"
    plan = queries.q3(*tabs)
for _ in range(5):
    plan.execute_device()
lib = ctx.lib
calls = []


def timed(name):
    fn = getattr(lib, name)

    def wrapper(*a):
        t0 = time.perf_counter()
        rc = fn(*a)
        calls.append((name, (time.perf_counter() - t0) * 1e6))
        return rc
    return wrapper


class Lib:
    def __getattr__(self, name):
        return timed(name) if name.endswith("_execute") else getattr(lib, name)


ctx.lib = Lib()
n = 50
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    before = ctx.sync_count()
    plan.execute_device()
waits = ctx.sync_count() - before
total = (time.perf_counter() - t0) / n * 1e6
per = {}
for name, us in calls:
    per.setdefault(name, []).append(us)
print(f"{sf}: {total:.0f} us per query, {waits} host wait(s)")
k = 0
for name, us in calls[-len(per):] if len(per) == 1 else calls[-3:]:
    print(f"  call {k} {name}: {us:.0f} us (last query)")
    k += 1
for name, v in per.items():
    print(f"  {name}: mean {sum(v) / len(v):.0f} us over {len(v)} calls")
print(f"  python around the calls: {total - sum(sum(v) for v in per.values()) / n:.0f} us")
