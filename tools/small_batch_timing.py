"""Upload and query time of Q3 at SF10 when the tables arrive as 1024-row batches (the reference's CSV loader's batch
size) vs the generator's big batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ctx = q.get_context()
c, o, l = synth.q3_tables(sf)


def rebatch(batches, rows):
    out = []
    for b in batches:
        out.extend(b.slice(k, min(rows, b.num_rows - k)) for k in range(0, b.num_rows, rows))
    return out


for name, (cc, oo, ll) in (("big batches", (c, o, l)), ("1024-row batches", (rebatch(c, 1024), rebatch(o, 1024), rebatch(l, 1024)))):
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, cc), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, oo), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, ll))
    nbytes = sum(b.nbytes for t in (cc, oo, ll) for b in t)
    t0 = time.perf_counter()
    for t in tabs:
        t.device_table()
    ctx.synchronize()
    up = time.perf_counter() - t0
    plan = queries.q3(*tabs)
    for _ in range(3):
        plan.execute_device()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        plan.execute_device()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 10
    j2 = plan.input
    t0 = time.perf_counter()
    out = j2.execute()
    dj = time.perf_counter() - t0
    print(f"{name}: {len(ll)} lineitem batches, upload {up:.3f} s ({nbytes / up / 1e9:.1f} GB/s), query {dt * 1e3:.3f} ms, "
          f"join-2 execute() incl. download of {len(out)} batches {dj * 1e3:.1f} ms", flush=True)
