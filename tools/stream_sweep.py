import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
ctx = q.get_context()
print(os.environ.get("QHIP_STREAM_UNROLL"), os.environ.get("QHIP_STREAM_NT"), os.environ.get("QHIP_STREAM_BLOCKS_PER_CU"), round(ctx.measure_stream_read(4 << 30, 10), 1), "GB/s", flush=True)
