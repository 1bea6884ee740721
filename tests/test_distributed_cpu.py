"""The N > 1 path on CPUs: world_size-2 `gloo` processes run the hash-join exchange (partition by key -> variable-size
all-to-all -> local join -> union) with the oracle as the local engine, and the union must equal the single-process join.
Covers qurious_amd.exchange.all_to_all_bytes (the same grouped send/recv code RCCL runs on GPUs) and the partitioning rule."""
import io
import os
import socket

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType

from .helpers import col, rows_of, table_scan

I64 = pa.int64()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tables(seed=21, nl=600, nr=2000, nkeys=97):
    rng = np.random.default_rng(seed)
    ls = pa.schema([pa.field("lk", I64), pa.field("lv", I64)])
    rs = pa.schema([pa.field("rk", I64), pa.field("rv", I64), pa.field("rs", pa.string())])
    lb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, nkeys, nl), type=I64, mask=rng.random(nl) < 0.05),
                                     pa.array(rng.integers(0, 10**6, nl), type=I64)], schema=ls)
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, nkeys, nr), type=I64, mask=rng.random(nr) < 0.05),
                                     pa.array(rng.integers(0, 10**6, nr), type=I64),
                                     pa.array(["s%d" % v for v in rng.integers(0, 9, nr)])], schema=rs)
    return (ls, lb), (rs, rb)


def _ipc(batch: pa.RecordBatch) -> bytes:
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue()


def _unipc(buf: bytes) -> pa.RecordBatch:
    return pa.ipc.open_stream(buf).read_all().combine_chunks().to_batches()[0]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd.exchange import all_to_all_bytes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (ls, lb), (rs, rb) = _tables()
        sides = []
        for schema, batch, key in ((ls, lb, "lk"), (rs, rb, "rk")):
            n = batch.num_rows
            mine = batch.slice(n * rank // world, n * (rank + 1) // world - n * rank // world)       # this rank's contiguous slice
            pid = qoracle.partition_ids([mine.column(key)], world)
            send = [torch.frombuffer(bytearray(_ipc(mine.filter(pa.array(pid == r)))), dtype=torch.uint8) for r in range(world)]
            got = all_to_all_bytes(send)
            parts = [_unipc(bytes(t.numpy().tobytes())) for t in got]
            sides.append((schema, pa.concat_batches(parts)))
        (ls2, lpart), (rs2, rpart) = sides
        # co-location: every key of this rank's inputs hashes to this rank
        for b, key in ((lpart, "lk"), (rpart, "rk")):
            assert (qoracle.partition_ids([b.column(key)], world) == rank).all()
        plan = q.HashJoinExec.try_new(table_scan(ls2, [lpart]), table_scan(rs2, [rpart]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
        local = rows_of(qoracle.execute(plan))
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            full = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
            want = sorted(rows_of(qoracle.execute(full)), key=repr)
            got_rows = sorted([r for part in gathered for r in part], key=repr)
            assert got_rows == want and len(want) > 1000
            open(os.path.join(out_dir, "ok"), "w").write(str(len(want)))
    finally:
        dist.destroy_process_group()


def test_exchange_join_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000


def test_partition_rule_is_deterministic_and_balanced():
    from oracle import qoracle
    keys = pa.array(np.arange(100000), type=I64)
    pid = qoracle.partition_ids([keys], 8)
    counts = np.bincount(pid, minlength=8)
    assert counts.min() > 11000 and counts.max() < 14000
    assert (qoracle.partition_ids([keys], 8) == pid).all()
    nulls = pa.array([None, None, 5], type=I64)
    p = qoracle.partition_ids([nulls], 4)
    assert p[0] == p[1]
