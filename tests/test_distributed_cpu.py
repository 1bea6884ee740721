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
            # metadata words ride with the sizes in the first round (the wire images' layout words on the GPU path)
            got = all_to_all_bytes(send, meta=[[rank, r, int(send[r].numel()) * 3] for r in range(world)])
            assert got.meta == [[r, rank, int(got[r].numel()) * 3] for r in range(world)]
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


def _agg_over(join_plan, aggs_of):
    """GROUP BY lk over a join's output: SUM / COUNT / MIN / MAX of rv"""
    schema = pa.schema([pa.field("lk", I64), pa.field("s", I64), pa.field("c", I64), pa.field("mn", I64), pa.field("mx", I64)])
    return q.HashAggregate(schema, join_plan, [col("lk", 0)], aggs_of(col("rv", 3))), schema


def _aggs(arg):
    return [q.SumAggregateExpr(arg, I64), q.CountAggregateExpr(arg), q.MinAggregateExpr(arg, I64), q.MaxAggregateExpr(arg, I64)]


def _worker_broadcast(rank, world, port, out_dir):
    """broadcast strategy: the build side is all-gathered, the probe slice stays local, partial groups are repartitioned
    by group key and merged (qurious_amd.exchange.BroadcastHashJoinExec / DistributedHashAggregate, with the oracle as the
    local engine)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd.exchange import all_to_all_bytes, merge_aggregate_exprs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (ls, lb), (rs, rb) = _tables(seed=33, nl=300, nr=4000, nkeys=150)
        cut = lambda b: b.slice(b.num_rows * rank // world, b.num_rows * (rank + 1) // world - b.num_rows * rank // world)   # noqa: E731
        lmine, rmine = cut(lb), cut(rb)
        # all-gather of the build side = an all-to-all that sends the same bytes to everybody
        payload = torch.frombuffer(bytearray(_ipc(lmine)), dtype=torch.uint8)
        build = pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes([payload] * world)])
        assert build.num_rows == lb.num_rows
        join = q.HashJoinExec.try_new(table_scan(ls, [build]), table_scan(rs, [rmine]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
        partial_plan, pschema = _agg_over(join, _aggs)
        partial = pa.concat_batches(qoracle.execute(partial_plan))
        pid = qoracle.partition_ids([partial.column(0)], world)
        send = [torch.frombuffer(bytearray(_ipc(partial.filter(pa.array(pid == r)))), dtype=torch.uint8) for r in range(world)]
        mine = pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes(send)])
        merge = q.HashAggregate(pschema, table_scan(pschema, [mine]), [col("lk", 0)], merge_aggregate_exprs(partial_plan.aggregate_exprs, 1))
        local = rows_of(qoracle.execute(merge))
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            full_join = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
            want = sorted(rows_of(qoracle.execute(_agg_over(full_join, _aggs)[0])))
            got = sorted(r for part in gathered for r in part)
            keys = [r[0] for r in got]
            assert len(keys) == len(set(keys))            # after the merge every group lives on exactly one rank
            assert got == want and len(want) > 50
            open(os.path.join(out_dir, "ok_broadcast"), "w").write(str(len(want)))
    finally:
        dist.destroy_process_group()


def test_broadcast_join_and_partial_aggregate_merge_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_broadcast, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok_broadcast").read()) > 50


def test_merge_rules_of_partial_aggregates():
    from qurious_amd.exchange import merge_aggregate_exprs
    merged = merge_aggregate_exprs(_aggs(col("rv", 3)), 2)
    assert [type(m).__name__ for m in merged] == ["SumAggregateExpr", "SumAggregateExpr", "MinAggregateExpr", "MaxAggregateExpr"]
    assert [m.expression().index for m in merged] == [2, 3, 4, 5]
    with pytest.raises(q.UnsupportedError, match="AVG"):
        merge_aggregate_exprs([q.AvgAggregateExpr(col("rv", 3), I64, pa.float64())], 1)


def test_exchange_join_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000


def test_exchange_join_world3_gloo(tmp_path):
    """an odd number of ranks: uneven slices, three-way grouped send / receive rounds"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
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


def test_exchange_column_pruning_analysis_on_q3():
    """prune_exchange_columns: every distributed join learns which output columns the plan above it reads; its exchanges
    carry those, its keys and nothing else (Q3: 4 of customer+orders' 7 columns, 3 of lineitem's 4)"""
    from qurious_amd import exchange, queries, synth
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, []), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, []),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, []))
    plan = queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)
    assert exchange.prune_exchange_columns(plan) is plan
    j2 = plan.partial.input
    j1 = j2.left
    names2 = [f.name for f in j2.schema()]
    assert sorted(names2[c] for c in j2._needed) == ["l_discount", "l_extendedprice", "l_orderkey", "o_orderdate", "o_shippriority"]
    l2, r2 = j2._needed_per_side()
    assert sorted(j1.schema().field(c).name for c in l2) == ["o_orderdate", "o_orderkey", "o_shippriority"]
    assert sorted(synth.LINEITEM_Q3_SCHEMA.field(c).name for c in r2) == ["l_discount", "l_extendedprice", "l_orderkey"]
    l1, r1 = j1._needed_per_side()
    assert [synth.CUSTOMER_SCHEMA.field(c).name for c in l1] == ["c_custkey"]           # c_mktsegment is only the scan filter's
    assert sorted(synth.ORDERS_SCHEMA.field(c).name for c in r1) == ["o_custkey", "o_orderdate", "o_orderkey", "o_shippriority"]
    wire = exchange._wire_schema(synth.CUSTOMER_SCHEMA, l1)
    assert [str(f.type) for f in wire] == ["int64", "null"]
    # an un-annotated plan keeps everything
    plain = queries.q3(*tabs, join_cls=exchange.DistributedHashJoinExec)
    assert plain.input._needed_per_side() == (None, None)
