"""Where the host -> HBM upload time of the 100 M-row lineitem table goes: first upload (cold device allocator) vs a
second one (pooled allocations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import synth
ctx = q.get_context()
t = time.perf_counter(); batches = synth.lineitem(int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 1 << 20); print("generate s", time.perf_counter() - t, flush=True)
nbytes = sum(b.nbytes for b in batches)
for it in range(3):
    t = time.perf_counter()
    dev = q.DeviceTable.from_batches(ctx, synth.LINEITEM_SCHEMA, batches)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"upload {it}: {dt:.3f} s = {nbytes / dt / 1e9:.1f} GB/s ({nbytes / 1e9:.2f} GB)", flush=True)
    del dev
