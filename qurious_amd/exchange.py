"""Multi-GPU hash-join exchange: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI; "gloo" on CPUs for tests) moves the bytes, libqhip does everything else on the device.

The reference has no exchange operator (it is a single process); this is the only place the path is partitioned:

    local slice of each join input --qhip_partition_by_key--> world parts (equal keys -> same part on every rank)
        --grouped send/recv (all-to-all; xGMI is point-to-point, so all 7 links of a GPU carry one peer each)-->
    parts received from every peer --qhip_table_from_device + qhip_table_concat--> local join input
        --qhip_hash_join_execute--> local slice of the join result

No collective is used for the aggregate after Q3's REPARTITIONED joins: its GROUP BY contains the join key, so groups are
disjoint across ranks and the result is the union of the ranks' results.

xGMI (7 links x ~153 GB/s) is ~30x slower per byte than HBM, so moving the big probe side is what limits scaling. The
second strategy keeps it where it is (SURVEY §8e's option):

    BroadcastHashJoinExec   the (small) build side of an Inner / Right join is ALL-GATHERED — every rank gets all of it —
                            and each rank probes it with its own slice of the probe side (never exchanged);
    DistributedHashAggregate  groups may then span ranks: every rank aggregates locally, the partial groups (few) are
                            repartitioned by group key and merged (SUM of sums, SUM of counts, MIN of mins, MAX of maxes).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional, Sequence, Tuple

from . import _ffi
from ._ffi import DeviceTable, get_context
from .datatypes import JoinType, to_qhip_dtype
from .expr import ExprArray, PhysicalExpr, int32_array
from .plan import HashJoinExec, PhysicalPlan


def _dist():
    import torch.distributed as dist
    return dist


def _exchange_world(dist) -> int:
    """Ranks to exchange with; 0 = run the plain single-process operator. QHIP_EXCHANGE_FORCE=1 keeps the exchange steps
    (partition, transport, rebuild, merge) in place for a ONE-rank group: the rehearsal that fits a one-GPU box."""
    import os
    if not dist.is_initialized():
        return 0
    world = dist.get_world_size()
    return world if world > 1 or os.environ.get("QHIP_EXCHANGE_FORCE") == "1" else 0


class qhip_comm_stats(C.Structure):
    _fields_ = [("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("bytes_packed", C.c_uint64), ("exchanges", C.c_uint64),
                ("host_waits", C.c_uint64), ("transfer_seconds", C.c_double), ("rank", C.c_int32), ("world", C.c_int32),
                ("rccl_version", C.c_int32), ("reserved", C.c_int32)]


def transport() -> str:
    """Who moves the wire images: "rccl" = libqhip's own communicator (qhip_exchange_tables / qhip_all_gather_table: RCCL
    dlopen'ed inside the library, everything on the context's stream, ONE host wait per exchange — what a torch-less host
    binds, include/qhip.h) or "torch" = torch.distributed point-to-point operations over device buffers. QHIP_TRANSPORT
    chooses; default: rccl when the process group runs on the nccl (= RCCL) backend, torch otherwise (gloo moves host memory)."""
    want = os.environ.get("QHIP_TRANSPORT", "")
    if want in ("rccl", "torch"):
        return want
    dist = _dist()
    return "rccl" if dist.is_initialized() and dist.get_backend() == "nccl" else "torch"


_COMM = {}


def get_comm(ctx=None):
    """libqhip's communicator over the ranks of torch.distributed's default group (created once per process): rank 0 draws
    the ncclUniqueId, the process group only carries its 128 bytes to the other ranks."""
    import torch
    ctx = ctx or get_context()
    if "h" in _COMM:
        return _COMM["h"]
    dist = _dist()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lib = ctx.lib
    lib.qhip_comm_unique_id.restype = C.c_int
    lib.qhip_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    lib.qhip_comm_create.restype = C.c_int
    lib.qhip_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.qhip_comm_get_stats.restype = C.c_int
    lib.qhip_comm_get_stats.argtypes = [C.c_void_p, C.POINTER(qhip_comm_stats), C.c_int32]
    lib.qhip_exchange_tables.restype = C.c_int
    lib.qhip_exchange_tables.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_char_p), C.POINTER(_ffi.qhip_dtype), C.c_int32,
                                         C.POINTER(C.c_void_p)]
    lib.qhip_all_gather_table.restype = C.c_int
    lib.qhip_all_gather_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(_ffi.qhip_dtype), C.c_int32,
                                          C.POINTER(C.c_void_p)]
    ident = (C.c_uint8 * 128)()
    if world > 1:
        if rank == 0:
            rc = lib.qhip_comm_unique_id(ident, 128)
            if rc != 0:
                _ffi._raise(rc, lib.qhip_last_error(None).decode())
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        ident = (C.c_uint8 * 128)(*t.cpu().tolist())
    h = C.c_void_p()
    ctx.check(lib.qhip_comm_create(ctx.handle, ident if world > 1 else None, rank, world, C.byref(h)))
    _COMM["h"], _COMM["ctx"] = h, ctx
    return h


def _schema_arrays(schema):
    ncols = len(schema)
    names = (C.c_char_p * max(1, ncols))(*[f.name.encode() for f in schema])
    dtypes = (_ffi.qhip_dtype * max(1, ncols))(*[to_qhip_dtype(f.type) for f in schema])
    return names, dtypes, ncols


def _comm_stats_into(reset: bool):
    """fold libqhip's communicator counters into _STATS (and clear them there)"""
    if "h" not in _COMM:
        return
    st = qhip_comm_stats()
    _COMM["ctx"].check(_COMM["ctx"].lib.qhip_comm_get_stats(_COMM["h"], C.byref(st), 1 if reset else 0))
    _STATS["bytes_sent"] += int(st.bytes_sent)
    _STATS["bytes_received"] += int(st.bytes_received)
    _STATS["bytes_packed"] += int(st.bytes_packed)
    _STATS["exchanges"] += int(st.exchanges)
    _STATS["seconds"] += float(st.transfer_seconds)
    _STATS["transport_waits"] = _STATS.get("transport_waits", 0) + int(st.host_waits)
    _STATS["rccl_version"] = int(st.rccl_version)


# bytes this rank handed to the transport and wall time spent inside the exchanges (BASELINE configs[3]/[4]: xGMI GB/s)
_STATS = {"bytes_sent": 0, "bytes_received": 0, "bytes_packed": 0, "seconds": 0.0, "exchanges": 0, "heavy_keys": 0, "probe_rows_received": 0,
          "transport_waits": 0, "heavy_key_rounds": 0, "range_rounds": 0, "build_rows_received": 0}


def exchange_stats(reset: bool = True) -> dict:
    """Totals since the last reset plus the derived GB/s per rank (to compare with 7 links x 153 GB/s of xGMI)."""
    _comm_stats_into(reset)
    out = dict(_STATS)
    out["send_GBps"] = out["bytes_sent"] / out["seconds"] / 1e9 if out["seconds"] > 0 else None
    if reset:
        _STATS.update(bytes_sent=0, bytes_received=0, bytes_packed=0, seconds=0.0, exchanges=0, heavy_keys=0, probe_rows_received=0,
                      transport_waits=0, heavy_key_rounds=0, range_rounds=0, build_rows_received=0)
    return out


class _Received(list):
    """the tensors received from every rank (a list), plus the metadata words that came with them (``.meta``)"""
    meta: list = None


def all_to_all_bytes(send: Sequence["torch.Tensor"], group=None, meta: Optional[Sequence[Sequence[int]]] = None) -> List["torch.Tensor"]:
    """Variable-size all-to-all of uint8 tensors: send[r] goes to rank r, the result's entry r came from rank r.
    ``meta[r]`` (equal-length int lists, optional) rides with the size in the first round; the result's ``.meta[r]`` is
    what rank r attached. Two rounds of grouped point-to-point sends/receives (``batch_isend_irecv``: one ncclGroup of
    ncclSend/ncclRecv on RCCL — xGMI is point-to-point, every peer has its own link — plain pairs on gloo): sizes +
    metadata, then payloads. Works for CPU tensors over gloo and GPU tensors over RCCL alike."""
    import time
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(send) == world
    dev = send[0].device
    # gloo moves host memory only: device tensors are staged through the host there (how several processes sharing ONE
    # GPU rehearse the exchange, tools/exchange_rehearsal.py --world 2); RCCL sends device memory as it is
    out_dev = dev
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        send = [t.cpu() for t in send]
        dev = torch.device("cpu")
    t_start = time.perf_counter()
    n_meta = len(meta[0]) if meta else 0
    head_out = torch.tensor([[send[r].numel()] + (list(meta[r]) if meta else []) for r in range(world)], dtype=torch.int64).to(dev)
    head_in = torch.zeros((world, 1 + n_meta), dtype=torch.int64, device=dev)
    head_in[rank].copy_(head_out[rank])
    ops = []
    for r in range(world):
        if r != rank:
            ops.append(dist.P2POp(dist.isend, head_out[r], r, group))
            ops.append(dist.P2POp(dist.irecv, head_in[r], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    heads = head_in.cpu().tolist()     # the one device -> host read of the exchange
    recv = _Received(torch.empty(int(heads[r][0]), dtype=torch.uint8, device=dev) for r in range(world))
    recv.meta = [h[1:] for h in heads]
    ops = []
    for r in range(world):
        if r == rank:
            recv[r].copy_(send[r])
            continue
        if send[r].numel():
            ops.append(dist.P2POp(dist.isend, send[r], r, group))
        if recv[r].numel():
            ops.append(dist.P2POp(dist.irecv, recv[r], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if out_dev != dev:
        staged = _Received(t.to(out_dev) for t in recv)
        staged.meta = recv.meta
        recv = staged
        torch.cuda.synchronize(out_dev)
    _STATS["bytes_sent"] += sum(int(t.numel()) for r, t in enumerate(send) if r != rank)
    _STATS["bytes_received"] += sum(int(t.numel()) for r, t in enumerate(recv) if r != rank)
    _STATS["seconds"] += time.perf_counter() - t_start
    _STATS["bytes_packed"] += sum(int(t.numel()) for t in send)     # incl. the part that stays on this rank
    _STATS["exchanges"] += 1
    _STATS["transport_waits"] = _STATS.get("transport_waits", 0) + (2 if dev.type == "cuda" else 0)   # the header read-back + the final synchronize
    return recv


class _DevMem:
    """Zero-copy view of device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _column_buffers(t: DeviceTable, col: int):
    """[(device pointer, bytes)] for values/offsets, validity, utf8 data of column `col`."""
    out = []
    for which in range(3):
        p, n = C.c_void_p(), C.c_int64()
        t.ctx.check(t.ctx.lib.qhip_table_column_buffer(t.handle, col, which, C.byref(p), C.byref(n)))
        out.append((p.value or 0, n.value))
    return out


class qhip_device_column(C.Structure):
    _fields_ = [("dtype", _ffi.qhip_dtype), ("length", C.c_int64), ("null_count", C.c_int64), ("values", C.c_void_p),
                ("validity", C.c_void_p), ("data", C.c_void_p), ("data_bytes", C.c_int64)]


def partition_by_key(table: DeviceTable, keys: Sequence[PhysicalExpr], n_parts: int) -> List[DeviceTable]:
    ctx = table.ctx
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    arr, n = ea.c_array()
    outs = (C.c_void_p * n_parts)()
    ctx.check(ctx.lib.qhip_partition_by_key(ctx.handle, table.handle, arr, n, int32_array(roots), len(roots), n_parts, outs))
    return [DeviceTable(ctx, C.c_void_p(outs[p])) for p in range(n_parts)]


def partition_filtered(table: DeviceTable, keys: Sequence[PhysicalExpr], n_parts: int, predicate: Optional[PhysicalExpr] = None,
                       keep: Optional[Sequence[bool]] = None, range_bounds: Optional[Sequence[int]] = None) -> List[DeviceTable]:
    """qhip_partition_filtered: the split by key hash fused with the join side's scan filter (`predicate`: rows it rejects are
    in no part) and with the projection pushdown of the exchange (`keep`: only these columns are moved, the others become
    NULL-typed placeholders) — two streaming passes over the columns involved, one host wait.
    ``range_bounds`` (n_parts - 1 ascending upper bounds of ONE integer key): by key range instead of by hash
    (qhip_partition_filtered_by_range)."""
    ctx = table.ctx
    ea = ExprArray()
    roots = [ea.lower(k) for k in keys]
    proot = ea.lower(predicate) if predicate is not None else -1
    arr, n = ea.c_array()
    outs = (C.c_void_p * n_parts)()
    keep_arr = None if keep is None else int32_array([1 if k else 0 for k in keep])
    if range_bounds is not None:
        assert len(range_bounds) == n_parts - 1
        bounds = (C.c_int64 * max(1, n_parts - 1))(*[int(b) for b in range_bounds])
        fn = ctx.lib.qhip_partition_filtered_by_range
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        ctx.check(fn(ctx.handle, table.handle, C.cast(arr, C.c_void_p), n, C.cast(int32_array(roots), C.c_void_p), len(roots), proot,
                     None if keep_arr is None else C.cast(keep_arr, C.c_void_p), C.cast(bounds, C.c_void_p), n_parts, C.cast(outs, C.c_void_p)))
    else:
        ctx.check(ctx.lib.qhip_partition_filtered(ctx.handle, table.handle, arr, n, int32_array(roots), len(roots), proot, keep_arr, n_parts, outs))
    return [DeviceTable(ctx, C.c_void_p(outs[p])) for p in range(n_parts)]


def column_range(table: DeviceTable, col: int) -> Tuple[int, int]:
    """qhip_table_column_range: [min, max] of an integer-like column (a deferred gather answers with its source's range)"""
    ctx = table.ctx
    fn = ctx.lib.qhip_table_column_range
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lo, hi = C.c_int64(), C.c_int64()
    ctx.check(fn(ctx.handle, table.handle, int(col), C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)


class qhip_shuffle_input(C.Structure):
    _fields_ = [("table", C.c_void_p), ("exprs", C.POINTER(_ffi.qhip_expr)), ("n_exprs", C.c_int32), ("key_roots", C.POINTER(C.c_int32)),
                ("n_keys", C.c_int32), ("predicate_root", C.c_int32), ("all_gather", C.c_int32), ("keep_columns", C.POINTER(C.c_int32)),
                ("range_bounds", C.POINTER(C.c_int64))]


def shuffle_tables(inputs) -> List[DeviceTable]:
    """qhip_shuffle_tables: the exchange step of a distributed join in ONE call with ONE host wait. `inputs`: tuples (device
    table, key expressions, scan filter or None, keep mask or None, all_gather) — the two sides of a repartitioned join, or
    the build side of a broadcast join. Raises UnsupportedError (on every rank alike) when a column cannot travel this way."""
    ctx = inputs[0][0].ctx
    lib = ctx.lib
    lib.qhip_shuffle_tables.restype = C.c_int
    lib.qhip_shuffle_tables.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(qhip_shuffle_input), C.c_int32, C.POINTER(C.c_void_p)]
    arr = (qhip_shuffle_input * len(inputs))()
    hold = []
    for k, item in enumerate(inputs):
        table, keys, predicate, keep, all_gather = item[:5]
        bounds = item[5] if len(item) > 5 else None     # world - 1 ascending upper bounds: rows go to ranks by key RANGE
        ea = ExprArray()
        roots = [ea.lower(e) for e in keys]
        proot = ea.lower(predicate) if predicate is not None else -1
        exprs, n = ea.c_array()
        roots_c = int32_array(roots)
        keep_c = None if keep is None else int32_array([1 if v else 0 for v in keep])
        bounds_c = None if bounds is None else (C.c_int64 * max(1, len(bounds)))(*[int(b) for b in bounds])
        hold.append((ea, exprs, roots_c, keep_c, bounds_c))
        arr[k].table = table.handle
        arr[k].exprs = exprs
        arr[k].n_exprs = n
        arr[k].key_roots = roots_c
        arr[k].n_keys = len(roots)
        arr[k].predicate_root = proot
        arr[k].all_gather = int(all_gather)   # bit 0: all-gather; bit 1: a join on the keys follows (their value ranges travel)
        arr[k].keep_columns = keep_c
        arr[k].range_bounds = bounds_c
    outs = (C.c_void_p * len(inputs))()
    ctx.check(lib.qhip_shuffle_tables(ctx.handle, get_comm(ctx), arr, len(inputs), outs))
    return [DeviceTable(ctx, C.c_void_p(outs[k])) for k in range(len(inputs))]


def _fast_exchange() -> bool:
    """the one-call exchange (qhip_shuffle_tables) needs libqhip's own communicator; QHIP_EXCHANGE_FAST=0 switches it off"""
    return _engine().fast_exchange and transport() == "rccl" and os.environ.get("QHIP_EXCHANGE_FAST", "1") != "0"


def _is_table_access(node) -> bool:
    from .plan import Scan
    return isinstance(node, Scan) and node.projections is None


def _side_for_exchange(node: PhysicalPlan, may_defer: bool):
    """(device table, scan filter or None) of a join side that goes into an exchange: a Scan's filter is handed to the exchange
    (evaluated in its first pass: the filtered batches are never materialised — valid for every join type, the rows are
    dropped BEFORE the join); any other node is executed, with a hash join at its top allowed to leave its size on the
    device when `may_defer` (the exchange reads it there, and every rank learns whether any rank must run it again)."""
    from .plan import _feeding
    if _is_table_access(node):
        return node.datasource.device_table(), node.filter
    if may_defer:
        return _feeding(get_context(), node), None
    return node.execute_device(), None


def concat_tables(tables: Sequence[DeviceTable]) -> DeviceTable:
    ctx = tables[0].ctx
    hs = (C.c_void_p * len(tables))(*[t.handle for t in tables])
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_table_concat(ctx.handle, hs, len(tables), C.byref(out)))
    return DeviceTable(ctx, out)


def _table_from_buffers(ctx, schema, rows: int, meta, bufs) -> DeviceTable:
    """meta[c] = (null_count, data_bytes); bufs[c] = [values, validity, data] uint8 device tensors"""
    ncols = len(schema)
    cols = (qhip_device_column * max(1, ncols))()
    names = (C.c_char_p * max(1, ncols))(*[f.name.encode() for f in schema])
    for c, f in enumerate(schema):
        cols[c].dtype = to_qhip_dtype(f.type)
        cols[c].length = rows
        cols[c].null_count = meta[c][0]
        cols[c].data_bytes = meta[c][1]
        v, n, d = bufs[c]
        cols[c].values = v.data_ptr() if v.numel() else None
        cols[c].validity = n.data_ptr() if n.numel() else None
        cols[c].data = d.data_ptr() if d.numel() else None
    out = C.c_void_p()
    fn = ctx.lib.qhip_table_from_device
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(qhip_device_column), C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]
    ctx.check(fn(ctx.handle, names, cols, ncols, rows, C.byref(out)))
    return DeviceTable(ctx, out)


def pack_table(table: DeviceTable):
    """(metadata words, uint8 device tensor): the wire image of `table` (qhip_table_pack, include/qhip.h)"""
    import torch
    ctx = table.ctx
    n_meta = 2 + 2 * table.num_columns
    m = (C.c_int64 * n_meta)()
    ctx.check(ctx.lib.qhip_table_wire_meta(ctx.handle, table.handle, m, n_meta))
    img = torch.empty(int(m[1]), dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    ctx.check(ctx.lib.qhip_table_pack(ctx.handle, table.handle, C.c_void_p(img.data_ptr() if img.numel() else None), img.numel()))
    return list(m), img


def unpack_concat(ctx, schema, metas, images) -> DeviceTable:
    """The concatenation, in the order given, of the tables the wire images describe (qhip_table_unpack_concat)."""
    ncols, n = len(schema), len(images)
    metas_c = (C.c_int64 * ((2 + 2 * ncols) * n))(*[int(v) for m in metas for v in m])
    ptrs = (C.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in images])
    names = (C.c_char_p * max(1, ncols))(*[f.name.encode() for f in schema])
    dtypes = (_ffi.qhip_dtype * max(1, ncols))(*[to_qhip_dtype(f.type) for f in schema])
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_table_unpack_concat(ctx.handle, names, dtypes, ncols, metas_c, ptrs, n, C.byref(out)))
    return DeviceTable(ctx, out)


def exchange_device_tables(parts: Sequence[DeviceTable], schema, group=None) -> DeviceTable:
    """parts[r] is sent to rank r; returns the concatenation (in rank order) of what every rank sent to this one.

    Every part travels as ONE wire image (all buffers of all columns in one allocation), so the exchange is two transport
    rounds whatever the schema: the images' metadata words (which double as the sizes), then the images. The received
    images are unpacked straight into the concatenated table."""
    dist = _dist()
    assert len(parts) == dist.get_world_size(group)
    if group is None and transport() == "rccl":
        ctx = parts[0].ctx
        names, dtypes, ncols = _schema_arrays(schema)
        hs = (C.c_void_p * len(parts))(*[p.handle for p in parts])
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_exchange_tables(ctx.handle, get_comm(ctx), hs, names, dtypes, ncols, C.byref(out)))
        return DeviceTable(ctx, out)
    E = _engine()
    packed = [E.pack(p) for p in parts]
    got = all_to_all_bytes([img for _, img in packed], group, meta=[m for m, _ in packed])
    return E.unpack(schema, got.meta, got)


def keep_columns(table: DeviceTable, keep: Optional[Sequence[bool]]) -> DeviceTable:
    """`table` with the columns not in `keep` replaced by NULL-typed placeholders (qhip_table_keep_columns); None = all"""
    if keep is None or all(keep):
        return table
    ctx = table.ctx
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_table_keep_columns(ctx.handle, table.handle, int32_array([1 if k else 0 for k in keep]), len(keep), C.byref(out)))
    return DeviceTable(ctx, out)


def referenced_columns(exprs: Sequence[PhysicalExpr]) -> set:
    """input-schema indices of the Column nodes inside the expressions"""
    from .expr import K_COLUMN
    ea = ExprArray()
    for e in exprs:
        ea.lower(e)
    return {int(n.column) for n in ea.nodes if n.kind == K_COLUMN}


def prune_exchange_columns(plan: PhysicalPlan, needed: Optional[set] = None) -> PhysicalPlan:
    """Projection pushdown through the exchanges: walk the plan from the root and tell every distributed join which of its
    output columns something above it reads (`needed`; None = all of them, as for the root). Its exchanges then move only
    those, the join keys and the residual filter's columns. Returns `plan` (annotated in place; plans stay valid without)."""
    from .plan import Filter, HashAggregate, Limit, Projection, Scan, Sort
    if isinstance(plan, DistributedHashAggregate):
        prune_exchange_columns(plan.partial, None)
    elif isinstance(plan, HashAggregate):
        prune_exchange_columns(plan.input, referenced_columns(list(plan.group_exprs) + [a.expression() for a in plan.aggregate_exprs]))
    elif isinstance(plan, Projection):
        prune_exchange_columns(plan.input, referenced_columns(plan.exprs))
    elif isinstance(plan, Filter):
        prune_exchange_columns(plan.input, None if needed is None else needed | referenced_columns([plan.predicate]))
    elif isinstance(plan, Sort):
        prune_exchange_columns(plan.input, None if needed is None else needed | referenced_columns([e.expr for e in plan.exprs]))
    elif isinstance(plan, Limit):
        prune_exchange_columns(plan.input, needed)
    elif isinstance(plan, HashJoinExec):
        plan._needed = None if needed is None else set(needed)
        sides = plan._needed_per_side()
        prune_exchange_columns(plan.left, sides[0])
        prune_exchange_columns(plan.right, sides[1])
    elif not isinstance(plan, Scan):
        for child in plan.children() or []:
            prune_exchange_columns(child, None)
    return plan


def _needed_per_side(join: HashJoinExec):
    """(left, right) input columns a join with `_needed` output columns reads: those, its keys, its residual filter's"""
    needed = getattr(join, "_needed", None)
    if needed is None:
        return None, None
    left = referenced_columns([l for l, _ in join.on])
    right = referenced_columns([r for _, r in join.on])
    for k, (index, side) in enumerate(join.column_indices):
        if k in needed:
            (left if int(side) == 0 else right).add(int(index))
    if join.filter is not None:
        for index, side in join.filter.column_indices:
            (left if int(side) == 0 else right).add(int(index))
    return left, right


HashJoinExec._needed_per_side = _needed_per_side


def _wire_schema(schema, needed: Optional[set]):
    """the schema the wire images are unpacked with: dropped columns are NULL-typed on both sides"""
    import pyarrow as pa
    if needed is None:
        return schema
    return pa.schema([f if c in needed else pa.field(f.name, pa.null()) for c, f in enumerate(schema)])


def _keep_mask(n_cols: int, needed: Optional[set]):
    return None if needed is None else [c in needed for c in range(n_cols)]


class LocalEngine:
    """What the multi-rank operators of this module ask of the RANK-LOCAL execution engine. The default — this class — is libqhip:
    tables are device tables in HBM and every method is a call through the C ABI. ``set_local_engine(other)`` installs another
    one; that is the documented hook through which tests/test_distributed_cpu.py drives DistributedHashJoinExec /
    BroadcastHashJoinExec / DistributedHashAggregate end to end on CPUs (gloo, world 2) with a pyarrow stub, so that the
    operators' RANK LOGIC — which side is exchanged, which columns travel, how heavy keys are split, how partial groups are
    merged — is covered where no GPU is. An engine's tables are opaque to the operators."""

    fast_exchange = True      # qhip_shuffle_tables (the one-call exchange) is available

    def context(self):
        return get_context()

    def execute(self, node: PhysicalPlan):
        return node.execute_device()

    def probe_side(self, join: HashJoinExec, fuse: bool):
        """(table, scan filter to fuse into the local join or None) of a join's probe side that stays on this rank"""
        return join._side(join.right, fuse)

    def base_table(self, scan):
        """the unfiltered table behind a Scan (heavy keys are sampled there)"""
        return scan.datasource.device_table()

    def num_rows(self, table) -> int:
        return table.num_rows

    def keep_columns(self, table, mask):
        return keep_columns(table, mask)

    def partition(self, table, keys, n_parts, range_bounds=None):
        return partition_filtered(table, keys, n_parts, range_bounds=range_bounds)

    def key_range(self, table, col: int):
        """(min, max) of an integer key column of a rank-local table (a superset is fine: it only steers the routing)"""
        return column_range(table, col)

    def key_share_in_range(self, table, schema, key, lo_excl, hi_incl) -> float:
        """fraction of a strided sample of `table`'s rows whose key lies in (lo_excl, hi_incl] (None = open): how much of a probe
        side would stay on this rank under range routing"""
        return _sample_share_in_range(table, schema, key, lo_excl, hi_incl)

    def pack(self, table):
        return pack_table(table)

    def unpack(self, schema, metas, images):
        return unpack_concat(get_context(), schema, metas, images)

    def concat(self, tables):
        return concat_tables(tables)

    def filter(self, table, predicate):
        from .plan import _filter_device
        return _filter_device(table, predicate, None)

    def top_keys(self, table, schema, key, dtype):
        return _local_top_keys(table, schema, key, dtype)

    def join(self, op: HashJoinExec, lt, rt, lpred=None, rpred=None):
        return op._join_tables(lt, rt, lpred, rpred)

    def aggregate(self, schema, table, keys, aggs):
        from .plan import HashAggregate
        return HashAggregate(schema, DeviceSource(schema, table), keys, aggs).execute_device()


_ENGINE = LocalEngine()


def set_local_engine(engine: Optional[LocalEngine]) -> LocalEngine:
    """Install the rank-local engine the multi-rank operators use (None: libqhip again); returns the previous one."""
    global _ENGINE
    previous = _ENGINE
    _ENGINE = engine if engine is not None else LocalEngine()
    return previous


def _engine() -> LocalEngine:
    return _ENGINE


class DeviceSource(PhysicalPlan):
    """A plan leaf over an already device-resident table (used to feed exchanged tables to HashJoinExec)."""

    def __init__(self, schema, table: DeviceTable):
        self._schema, self.table = schema, table

    def schema(self):
        return self._schema

    def execute_device(self) -> DeviceTable:
        return self.table


# ---------------------------------------------------------------- heavy hitters of a repartitioned join
# Repartitioning by key hash sends every row of one key to one rank: with skewed probe keys (BASELINE configs[4]: Zipf 1.1)
# a handful of keys would give one rank most of the probe side. Keys holding more than 1 / (4 n) of the probe rows are
# therefore found on a SAMPLE first; their probe rows stay where they are and their (few) build rows are broadcast.
HEAVY_SAMPLE_STRIDE = 64      # every 64th probe row is counted
HEAVY_CANDIDATES = 64         # per rank: its most frequent sampled keys (a globally heavy key is among the top 4 n <= 32 somewhere)
HEAVY_REFRESH = 16            # executions of one join plan between two samplings of its probe side


def heavy_keys(local_top: Sequence[Tuple[object, int]], local_sample_rows: int, group=None) -> List[object]:
    """The globally heavy keys, identical on every rank: every rank contributes its most frequent sampled keys with their
    counts and its sample size; a key is heavy when the sum of its reported counts exceeds total sample / (4 world). (A key
    not among some rank's candidates is under-counted there by less than that rank's 1/HEAVY_CANDIDATES share: only keys
    right at the threshold can be missed, and missing one costs balance only.)
    Integer keys travel as ONE fixed-shape int64 tensor all-gather ([sample rows | HEAVY_CANDIDATES x (key, count)]: no
    pickling, no size round); other key types (strings) fall back to an all-gather of Python objects."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    local_top = [(k, int(c)) for k, c in local_top if k is not None][:HEAVY_CANDIDATES]
    ints = all(isinstance(k, int) and -(1 << 63) <= k < (1 << 63) for k, _ in local_top)
    row = [int(local_sample_rows)]
    for k, c in (local_top if ints else []):
        row += [int(k), int(c)]
    row += [0, 0] * (HEAVY_CANDIDATES - (len(local_top) if ints else 0))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    if dev.type == "cuda" and "h" in _COMM:
        # libqhip's own communicator may still have a grouped send / recv in flight on the context's stream: two RCCL
        # communicators active at once on one device is the hazard NCCL warns about — drain ours before torch's runs (ADVICE r03;
        # this sampling round happens once in HEAVY_REFRESH executions)
        _COMM["ctx"].synchronize()
    mine = torch.tensor(row + [1 if ints else 0], dtype=torch.int64, device=dev)     # last word: "my keys are integers"
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine, group=group)                                      # (every rank, whatever its key type)
    rows = [t.cpu().tolist() for t in everyone]
    if all(r[-1] == 1 for r in rows):
        gathered = [([(r[1 + 2 * j], r[2 + 2 * j]) for j in range(HEAVY_CANDIDATES) if r[2 + 2 * j] > 0], r[0]) for r in rows]
    else:   # some rank holds non-integer keys: every rank saw that in the flag words and takes the object path
        gathered = [None] * world
        dist.all_gather_object(gathered, (list(local_top), int(local_sample_rows)), group=group)
    _STATS["heavy_key_rounds"] = _STATS.get("heavy_key_rounds", 0) + 1
    total = sum(n for _, n in gathered)
    counts = {}
    for top, _ in gathered:
        for key, c in top:
            if key is not None:
                counts[key] = counts.get(key, 0) + int(c)
    limit = total / (4.0 * world)
    return sorted(k for k, c in counts.items() if c > limit)


def _key_literal(value, dtype):
    import pyarrow as pa
    from .datatypes import ScalarValue
    from .expr import CastExpr, Literal
    if pa.types.is_int64(dtype):
        return Literal(ScalarValue.Int64(int(value)))
    if pa.types.is_int32(dtype):
        return Literal(ScalarValue.Int32(int(value)))
    if pa.types.is_date32(dtype):
        return CastExpr(Literal(ScalarValue.Int32(int(value))), pa.date32())
    if pa.types.is_string(dtype):
        return Literal(ScalarValue.Utf8(str(value)))
    return None


def heavy_split_predicates(key: PhysicalExpr, dtype, keys: Sequence[object]):
    """(is-heavy, is-not-heavy) predicates over one key expression. A NULL key is not heavy: it takes the ordinary path
    (IS NULL OR (key <> h1 AND key <> h2 ...)), so the two predicates split a table exactly."""
    from .datatypes import Operator
    from .expr import BinaryExpr, IsNull
    lits = [_key_literal(k, dtype) for k in keys]
    if not lits or any(l is None for l in lits):
        return None, None
    heavy, light = None, None
    for l in lits:
        eq, ne = BinaryExpr(key, Operator.Eq, l), BinaryExpr(key, Operator.NotEq, l)
        heavy = eq if heavy is None else BinaryExpr(heavy, Operator.Or, eq)
        light = ne if light is None else BinaryExpr(light, Operator.And, ne)
    return heavy, BinaryExpr(IsNull(key), Operator.Or, light)


def _local_top_keys(table: DeviceTable, schema, key: PhysicalExpr, dtype):
    """[(key value, count)] of the most frequent keys among every HEAVY_SAMPLE_STRIDE-th row of `table`, and the sample size:
    a GROUP BY key / COUNT over the sample, top-N by count — all on the device, HEAVY_CANDIDATES rows come back."""
    import pyarrow as pa
    from .expr import Column, CountAggregateExpr, Literal
    from .datatypes import ScalarValue
    from .plan import HashAggregate, Limit, PhysicalSortExpr, Sort, SortOptions
    ctx = table.ctx
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_table_stride_sample(ctx.handle, table.handle, HEAVY_SAMPLE_STRIDE, C.byref(out)))
    sample = DeviceTable(ctx, out)
    if sample.num_rows == 0:
        return [], 0
    gschema = pa.schema([pa.field("k", dtype), pa.field("n", pa.int64())])
    agg = HashAggregate(gschema, DeviceSource(schema, sample), [key], [CountAggregateExpr(Literal(ScalarValue.Int64(1)))])
    top = Limit(Sort([PhysicalSortExpr(Column("n", 1), SortOptions(True, False))], agg, HEAVY_CANDIDATES), HEAVY_CANDIDATES, 0)
    rows = []
    for b in top.execute():
        rows.extend(zip(b.column(0).to_pylist(), b.column(1).to_pylist()))
    return [(k, c) for k, c in rows if k is not None], sample.num_rows


def _sample_share_in_range(table: DeviceTable, schema, key: PhysicalExpr, lo_excl, hi_incl) -> float:
    from .datatypes import Operator, ScalarValue
    from .expr import BinaryExpr, CountAggregateExpr, Literal
    from .plan import Filter, NoGroupingAggregate
    import pyarrow as pa
    ctx = table.ctx
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_table_stride_sample(ctx.handle, table.handle, HEAVY_SAMPLE_STRIDE, C.byref(out)))
    sample = DeviceTable(ctx, out)
    if sample.num_rows == 0:
        return 1.0
    dtype = _expr_type(key, schema)
    pred = None
    for bound, op in ((lo_excl, Operator.Gt), (hi_incl, Operator.LtEq)):
        if bound is None:
            continue
        lit = _key_literal(bound, dtype)
        if lit is None:
            return 0.0
        term = BinaryExpr(key, op, lit)
        pred = term if pred is None else BinaryExpr(pred, Operator.And, term)
    src = DeviceSource(schema, sample)
    node = src if pred is None else Filter(src, pred)
    agg = NoGroupingAggregate(pa.schema([pa.field("n", pa.int64())]), node, [CountAggregateExpr(Literal(ScalarValue.Int64(1)))])
    n = sum(b.column(0).to_pylist()[0] for b in agg.execute())
    return float(n) / float(sample.num_rows)


class DistributedHashJoinExec(HashJoinExec):
    """HashJoinExec whose inputs are this rank's slices: both sides are repartitioned by the join key across the ranks of
    ``torch.distributed``'s default group, then joined locally. Row order across ranks is not the single-process order
    (the north star asks for row-SET equality); inside a rank the reference's order holds."""

    @property
    def _exchanges(self) -> bool:
        """plan.py `_feeding`: a subtree with an active exchange operator never runs joins of deferred size"""
        return bool(_exchange_world(_dist()))

    def execute_device(self) -> DeviceTable:
        from .plan import _retrying, exchange_cache
        world = _exchange_world(_dist())
        if not world:
            return HashJoinExec.execute_device(self)
        ctx = _engine().context()

        def once():
            # what this operator received earlier in the SAME execution of the plan (a local retry above it): no collective again
            got = exchange_cache(ctx).get(id(self))
            if got is None:
                got = self._exchange_inputs(world)
                exchange_cache(ctx)[id(self)] = got
            return _engine().join(self, *got)
        return _retrying(ctx, once)

    def _heavy_keys_now(self, world, rs, rw):
        """the join's heavy probe keys (cached; sampled on the probe side's base table every HEAVY_REFRESH executions), or None
        when heavy-hitter handling does not apply to this join"""
        if len(self.on) != 1 or self.join_type not in (JoinType.Inner, JoinType.Right) or os.environ.get("QHIP_EXCHANGE_NO_HEAVY") == "1":
            return None
        rkey = self.on[0][1]
        rdtype = _expr_type(rkey, rs)
        if rdtype is None or not _is_table_access(self.right):
            return None
        cached = getattr(self, "_heavy_cache", None)
        if cached is not None and cached[0] == world and cached[2] > 0 and os.environ.get("QHIP_EXCHANGE_NO_HEAVY_CACHE") != "1":
            self._heavy_cache = (world, cached[1], cached[2] - 1)
            return cached[1]
        # (sampled on the UNFILTERED table: a heavy key of the table is what matters for balance; a stale or approximate set
        # costs balance, never correctness)
        top, n_sample = _engine().top_keys(_engine().base_table(self.right), rs, rkey, rdtype)
        keys = heavy_keys(top, n_sample)
        self._heavy_cache = (world, keys, HEAVY_REFRESH - 1)
        return keys

    def _range_bounds_now(self, world, build_table, probe_table):
        """world - 1 ascending upper bounds when the rows are to be routed by KEY RANGE instead of by key hash, else None.
        QHIP_EXCHANGE_RANGE=1 (off by default: DESIGN §7 "routing by key range"). Every rank contributes the [min, max] of its
        build side's key column (a statistic of the resident table, computed once); when the ranks' ranges are disjoint and
        ascending with the rank — tables sliced in key order, TPC-H's orders and lineitem — rank r gets the keys up to ITS maximum.
        BOTH sides of the join are split by the same bounds, so the join is correct whatever the bounds are: a stale or
        lopsided set costs balance or traffic only. It is taken only when the PROBE side is laid out the same way (a rank's probe
        keys reach no further than its neighbours' build ranges): routing unordered or skewed probe rows by range would give up
        the hash's balance and the heavy-hitter handling. The agreed bounds are kept and refreshed every HEAVY_REFRESH executions."""
        if os.environ.get("QHIP_EXCHANGE_RANGE", "0") != "1" or len(self.on) != 1 or world < 2:
            return None
        import pyarrow as pa
        from .expr import Column
        lkey, rkey = self.on[0]
        if not (isinstance(lkey, Column) and isinstance(rkey, Column)):
            return None
        lt_, rt_ = _expr_type(lkey, self.left.schema()), _expr_type(rkey, self.right.schema())
        ok = lambda t: t is not None and (pa.types.is_integer(t) or pa.types.is_date(t))   # noqa: E731
        if not (ok(lt_) and ok(rt_)):
            return None
        cached = getattr(self, "_range_cache", None)
        if cached is not None and cached[0] == world and cached[2] > 0:
            self._range_cache = (world, cached[1], cached[2] - 1)
            return cached[1]
        import torch
        dist = _dist()
        rank = dist.get_rank()
        lo, hi = _engine().key_range(build_table, lkey.index)
        plo, phi = _engine().key_range(probe_table, rkey.index)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        if dev.type == "cuda" and "h" in _COMM:
            _COMM["ctx"].synchronize()   # (two RCCL communicators never active at once: see heavy_keys)
        mine = torch.tensor([int(lo), int(hi), int(plo), int(phi)], dtype=torch.int64, device=dev)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        ranges = [tuple(t.cpu().tolist()) for t in everyone]
        bounds = [r[1] for r in ranges[:-1]]
        disjoint = all(ranges[r][0] <= ranges[r][1] for r in range(world)) and all(ranges[r][1] < ranges[r + 1][0] for r in range(world - 1))
        # the probe slices follow the build slices: rank r's probe keys lie between the build ranges of ranks r - 1 and r + 1
        aligned = all((r < 2 or ranges[r][2] > ranges[r - 2][1]) and (r + 2 >= world or ranges[r][3] < ranges[r + 2][0]) for r in range(world)
                      if ranges[r][2] <= ranges[r][3])
        bounds = bounds if (disjoint and aligned) else None
        if bounds is not None:
            # ... and most of every rank's probe rows would stay where they are (a strided sample): unordered or skewed probe keys
            # inside aligned RANGES (Zipf-drawn foreign keys) keep the hash's balance and heavy-hitter handling
            share = _engine().key_share_in_range(probe_table, self.right.schema(), rkey, bounds[rank - 1] if rank > 0 else None,
                                                 bounds[rank] if rank < world - 1 else None)
            mine = torch.tensor([int(share * 1000)], dtype=torch.int64, device=dev)
            everyone = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(everyone, mine)
            if min(int(t.cpu().item()) for t in everyone) < 800:   # (co-located tables: > 0.95; keys drawn at random: 1 / world)
                bounds = None
        _STATS["range_rounds"] = _STATS.get("range_rounds", 0) + 1
        self._range_cache = (world, bounds, HEAVY_REFRESH - 1)
        return bounds

    def _exchange_inputs(self, world):
        """(build table, probe table) of the local join: both sides repartitioned by key hash — or by key range"""
        ls, rs = self.left.schema(), self.right.schema()
        lneed, rneed = self._needed_per_side()
        if _fast_exchange():
            lt, lpred = _side_for_exchange(self.left, _is_table_access(self.right))
            rt, rpred = _side_for_exchange(self.right, False)
            bounds = self._range_bounds_now(world, lt, rt)
            heavy = None if bounds is not None else self._heavy_keys_now(world, rs, _wire_schema(rs, rneed))
            if not heavy:
                # ONE call, ONE host wait for both sides; the build side may be a join of deferred size when nothing executes
                # between it and the exchange (the probe side is a table access)
                try:
                    got = shuffle_tables([(lt, [l for l, _ in self.on], lpred, _keep_mask(len(ls), lneed), 2, bounds),
                                          (rt, [r for _, r in self.on], rpred, _keep_mask(len(rs), rneed), 2, bounds)])
                    _STATS["probe_rows_received"] = _STATS.get("probe_rows_received", 0) + got[1].num_rows
                    return got[0], got[1]
                except _ffi.UnsupportedError:
                    pass   # (a column kind the fast path does not move — every rank decides alike: the generic path below)
        ctx = _engine().context()
        if not hasattr(ctx, "no_deferred_sizes"):
            return self._execute_exchanged(world)
        with ctx.no_deferred_sizes():   # (the generic path settles its inputs one by one: no sizes left on the device)
            return self._execute_exchanged(world)

    def _execute_exchanged(self, world):
        E = _engine()
        ls, rs = self.left.schema(), self.right.schema()
        lneed, rneed = self._needed_per_side()
        # columns nothing above this join reads are dropped BEFORE the partitioning: never gathered, never sent
        left = E.keep_columns(E.execute(self.left), _keep_mask(len(ls), lneed))
        right = E.keep_columns(E.execute(self.right), _keep_mask(len(rs), rneed))
        lw, rw = _wire_schema(ls, lneed), _wire_schema(rs, rneed)
        bounds = self._range_bounds_now(world, left, right)
        if bounds is not None:   # by key range: both sides split by the same bounds, no heavy-hitter handling (co-located rows stay)
            lt = exchange_device_tables(E.partition(left, [l for l, _ in self.on], world, bounds), lw)
            rt = exchange_device_tables(E.partition(right, [r for _, r in self.on], world, bounds), rw)
            _STATS["probe_rows_received"] = _STATS.get("probe_rows_received", 0) + E.num_rows(rt)
            return lt, rt
        heavy_l = heavy_r = None
        # heavy hitters (one key column, join types in which a result row belongs to exactly one probe row): their probe rows
        # stay on this rank, their build rows go to every rank; everything else is repartitioned by key hash
        if len(self.on) == 1 and self.join_type in (JoinType.Inner, JoinType.Right) and os.environ.get("QHIP_EXCHANGE_NO_HEAVY") != "1":
            lkey, rkey = self.on[0]
            rdtype = _expr_type(rkey, rs)
            if rdtype is not None:
                # the heavy-key set is a learnt property of the join like its pair counts: found on the first execution and
                # kept, refreshed every HEAVY_REFRESH executions (all ranks count executions alike, so they refresh together);
                # a stale set costs balance, never correctness (the two predicates split the rows exactly whatever the keys)
                cached = getattr(self, "_heavy_cache", None)
                if cached is not None and cached[0] == world and cached[2] > 0 and os.environ.get("QHIP_EXCHANGE_NO_HEAVY_CACHE") != "1":
                    keys = cached[1]
                    self._heavy_cache = (world, keys, cached[2] - 1)
                else:
                    top, n_sample = E.top_keys(right, rw, rkey, rdtype)
                    keys = heavy_keys(top, n_sample)
                    self._heavy_cache = (world, keys, HEAVY_REFRESH - 1)
                hl, ll = heavy_split_predicates(lkey, _expr_type(lkey, ls), keys) if keys else (None, None)
                hr, lr = heavy_split_predicates(rkey, rdtype, keys) if keys else (None, None)
                if hl is not None and hr is not None:
                    heavy_l, left = all_gather_device_table(E.filter(left, hl), lw), E.filter(left, ll)
                    heavy_r, right = E.filter(right, hr), E.filter(right, lr)
                    _STATS["heavy_keys"] = _STATS.get("heavy_keys", 0) + len(keys)
        lt = exchange_device_tables(E.partition(left, [l for l, _ in self.on], world), lw)
        rt = exchange_device_tables(E.partition(right, [r for _, r in self.on], world), rw)
        _STATS["probe_rows_received"] = _STATS.get("probe_rows_received", 0) + E.num_rows(rt) + (E.num_rows(heavy_r) if heavy_r is not None else 0)
        if heavy_l is not None:
            lt, rt = E.concat([lt, heavy_l]), E.concat([rt, heavy_r])
        return lt, rt

    @staticmethod
    def try_new(left, right, join_type, on, filter=None) -> "DistributedHashJoinExec":
        base = HashJoinExec.try_new(left, right, JoinType(join_type), on, filter)
        return DistributedHashJoinExec(base.left, base.right, base.join_type, base.on, base.filter, base._schema, base.column_indices)


def _expr_type(e: PhysicalExpr, schema):
    """Arrow type of a key expression that is a plain column (what heavy-hitter handling supports), else None"""
    from .expr import Column
    return schema.field(e.index).type if isinstance(e, Column) and 0 <= e.index < len(schema) else None


def all_gather_device_table(table: DeviceTable, schema, group=None) -> DeviceTable:
    """Every rank ends up with the concatenation (rank order) of all ranks' tables."""
    world = _dist().get_world_size(group)
    if group is None and transport() == "rccl":
        ctx = table.ctx
        names, dtypes, ncols = _schema_arrays(schema)
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_all_gather_table(ctx.handle, get_comm(ctx), table.handle, names, dtypes, ncols, C.byref(out)))
        return DeviceTable(ctx, out)
    E = _engine()
    meta, img = E.pack(table)              # packed once, the same image goes to every peer
    got = all_to_all_bytes([img] * world, group, meta=[meta] * world)
    return E.unpack(schema, got.meta, got)


class BroadcastHashJoinExec(HashJoinExec):
    """HashJoinExec for a small build side: it is replicated with one all-gather and every rank joins it with its local
    slice of the probe side, which never moves. Valid for the join types in which a result row belongs to exactly one
    probe row or pair (Inner, Right); the others fall back to the repartitioned join. Inside a rank the reference's
    output order holds; across ranks the result is the union."""

    _exchanges = DistributedHashJoinExec._exchanges

    def execute_device(self) -> DeviceTable:
        from .plan import _retrying, exchange_cache
        world = _exchange_world(_dist())
        if not world:
            return HashJoinExec.execute_device(self)
        if self.join_type not in (JoinType.Inner, JoinType.Right):
            return DistributedHashJoinExec.execute_device(self)
        ctx = _engine().context()

        def once():
            build = exchange_cache(ctx).get(id(self))
            if build is None:
                build = self._gather_build()
                exchange_cache(ctx)[id(self)] = build
            probe, rpred = _engine().probe_side(self, self.join_type == JoinType.Inner)
            return _engine().join(self, build, probe, None, rpred)
        return _retrying(ctx, once)

    def _gather_build(self) -> DeviceTable:
        """every rank's build rows on every rank: ONE call, ONE host wait, the build side's scan filter evaluated in its first
        pass and a build side that is a join of deferred size read as it is (qhip_shuffle_tables, all_gather)"""
        ls = self.left.schema()
        lneed, _ = self._needed_per_side()
        if _fast_exchange():
            try:
                lt, lpred = _side_for_exchange(self.left, _is_table_access(self.right))
                return shuffle_tables([(lt, [l for l, _ in self.on], lpred, _keep_mask(len(ls), lneed), 3)])[0]
            except _ffi.UnsupportedError:
                pass
        E = _engine()
        ctx = E.context()
        if not hasattr(ctx, "no_deferred_sizes"):
            return all_gather_device_table(E.keep_columns(E.execute(self.left), _keep_mask(len(ls), lneed)), _wire_schema(ls, lneed))
        with ctx.no_deferred_sizes():
            return all_gather_device_table(E.keep_columns(E.execute(self.left), _keep_mask(len(ls), lneed)), _wire_schema(ls, lneed))

    _exchange_inputs = DistributedHashJoinExec._exchange_inputs
    _execute_exchanged = DistributedHashJoinExec._execute_exchanged
    _heavy_keys_now = DistributedHashJoinExec._heavy_keys_now

    @staticmethod
    def try_new(left, right, join_type, on, filter=None) -> "BroadcastHashJoinExec":
        base = HashJoinExec.try_new(left, right, JoinType(join_type), on, filter)
        return BroadcastHashJoinExec(base.left, base.right, base.join_type, base.on, base.filter, base._schema, base.column_indices)


class RangeBroadcastHashJoinExec(BroadcastHashJoinExec):
    """BroadcastHashJoinExec that sends a build row only to the ranks whose PROBE keys can match it (round 4, DESIGN §7 "routing
    by key range"; QHIP_EXCHANGE_RANGE=1, else exactly BroadcastHashJoinExec). Every rank all-gathers the [min, max] of its probe
    side's key column (a statistic of the resident table); a rank then receives the build rows whose key lies in ITS probe range —
    which is all an Inner / Right join of its probe rows can ever match, whatever the ranges are — and joins them, together with
    its own build rows, with its probe side IN PLACE (fused scan filter, nothing of the big side is copied or moved). With tables
    sliced in key order (TPC-H's orders and lineitem) a rank's probe range is its neighbours' border at most: the exchange carries
    a few rows and every rank does 1/n of the single-GPU work: ONE range partition of the small side (a row goes to the rank whose
    probe range ends at or above its key), the keys at which two neighbours' ranges overlap sent to the next rank as well, one
    all-to-all. When the ranks' probe ranges do not ascend with the rank or reach beyond their direct neighbours (unordered keys):
    plain broadcast. The ranges are gathered at EVERY execution (a stale range would lose
    matches, and a refresh must be collective)."""

    def _probe_ranges(self, world):
        if os.environ.get("QHIP_EXCHANGE_RANGE", "0") != "1" or len(self.on) != 1 or world < 2 or not _is_table_access(self.right):
            return None
        import pyarrow as pa
        import torch
        from .expr import Column
        lkey, rkey = self.on[0]
        if not (isinstance(lkey, Column) and isinstance(rkey, Column)):
            return None
        lt_, rt_ = _expr_type(lkey, self.left.schema()), _expr_type(rkey, self.right.schema())
        ok = lambda t: t is not None and (pa.types.is_int64(t) or pa.types.is_int32(t) or pa.types.is_date32(t))   # noqa: E731
        if not (ok(lt_) and ok(rt_)):
            return None
        E = _engine()
        dist = _dist()
        plo, phi = E.key_range(E.base_table(self.right), rkey.index)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        if dev.type == "cuda" and "h" in _COMM:
            _COMM["ctx"].synchronize()   # (two RCCL communicators never active at once: see heavy_keys)
        mine = torch.tensor([int(plo), int(phi)], dtype=torch.int64, device=dev)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        ranges = [tuple(t.cpu().tolist()) for t in everyone]
        _STATS["range_rounds"] = _STATS.get("range_rounds", 0) + 1
        live = [r for r in ranges if r[0] <= r[1]]
        if len(live) < world:
            return None        # (a rank without probe rows: nothing to gain, the plain broadcast covers it)
        # the ranks' probe ranges ascend with the rank and reach at most into their direct neighbour's (tables sliced in key order)
        ascending = all(ranges[r][0] <= ranges[r + 1][0] and ranges[r][1] <= ranges[r + 1][1] for r in range(world - 1))
        shallow = all(ranges[r + 2][0] > ranges[r][1] for r in range(world - 2))
        # ... and overlap in a small part of the key space only (what overlaps is sent twice)
        span = ranges[-1][1] - ranges[0][0] + 1
        overlap = sum(max(0, ranges[r][1] - ranges[r + 1][0] + 1) for r in range(world - 1))
        return ranges if (ascending and shallow and overlap * 10 <= span) else None

    def _gather_build(self) -> DeviceTable:
        world = _exchange_world(_dist())
        ranges = self._probe_ranges(world) if self.join_type in (JoinType.Inner, JoinType.Right) else None
        if ranges is None:
            return BroadcastHashJoinExec._gather_build(self)
        from .datatypes import Operator
        from .expr import BinaryExpr
        E = _engine()
        ctx = E.context()
        ls = self.left.schema()
        lneed, _ = self._needed_per_side()
        lkey = self.on[0][0]
        dtype = _expr_type(lkey, ls)
        uppers = [r[1] for r in ranges[:-1]]          # part r = the keys in (max of rank r - 1, max of rank r]; the last part the rest

        def run():
            build = E.keep_columns(E.execute(self.left), _keep_mask(len(ls), lneed))
            # ONE range partition of the (small) build side: a row goes to the rank whose probe range ends at or above its key ...
            parts = E.partition(build, [lkey], world, uppers)
            # ... and a key at which two neighbours' probe ranges OVERLAP (an order whose lineitems straddle the slice border) to
            # the next rank as well: those few rows are found by one Filter and split by the same bounds
            overlap = None
            for r in range(world - 1):
                lo, hi = ranges[r + 1][0], ranges[r][1]
                if lo <= hi:
                    term = BinaryExpr(BinaryExpr(lkey, Operator.GtEq, _key_literal(lo, dtype)), Operator.And, BinaryExpr(lkey, Operator.LtEq, _key_literal(hi, dtype)))
                    overlap = term if overlap is None else BinaryExpr(overlap, Operator.Or, term)
            if overlap is not None:
                again = E.partition(E.filter(build, overlap), [lkey], world, uppers)
                parts = [parts[d] if d == 0 else E.concat([parts[d], again[d - 1]]) for d in range(world)]
            got = exchange_device_tables(parts, _wire_schema(ls, lneed))
            _STATS["build_rows_received"] = _STATS.get("build_rows_received", 0) + E.num_rows(got) - E.num_rows(parts[_dist().get_rank()])
            return got
        if not hasattr(ctx, "no_deferred_sizes"):
            return run()
        with ctx.no_deferred_sizes():
            return run()

    @staticmethod
    def try_new(left, right, join_type, on, filter=None) -> "RangeBroadcastHashJoinExec":
        base = HashJoinExec.try_new(left, right, JoinType(join_type), on, filter)
        return RangeBroadcastHashJoinExec(base.left, base.right, base.join_type, base.on, base.filter, base._schema, base.column_indices)


def merge_aggregate_exprs(aggregate_exprs, n_groups: int):
    """The aggregate list that merges partial results laid out as (group keys..., partials...): SUM -> SUM of the partial
    sums, COUNT -> SUM of the partial counts, MIN / MAX -> MIN / MAX of the partials. AVG has no single-column partial."""
    import pyarrow as pa
    from .expr import AvgAggregateExpr, Column, CountAggregateExpr, MaxAggregateExpr, MinAggregateExpr, SumAggregateExpr
    out = []
    for k, a in enumerate(aggregate_exprs):
        part = Column(f"partial{k}", n_groups + k)
        if isinstance(a, AvgAggregateExpr):
            raise _ffi.UnsupportedError(_ffi.QHIP_UNSUPPORTED, "AVG cannot be merged from single-column partials: aggregate SUM and COUNT instead")
        if isinstance(a, CountAggregateExpr):
            out.append(SumAggregateExpr(part, pa.int64()))
        elif isinstance(a, SumAggregateExpr):
            out.append(SumAggregateExpr(part, a._return_type()))
        elif isinstance(a, MinAggregateExpr):
            out.append(MinAggregateExpr(part, a._return_type()))
        elif isinstance(a, MaxAggregateExpr):
            out.append(MaxAggregateExpr(part, a._return_type()))
        else:
            raise _ffi.UnsupportedError(_ffi.QHIP_UNSUPPORTED, f"no merge rule for {type(a).__name__}")
    return out


class DistributedHashAggregate(PhysicalPlan):
    """HashAggregate whose input rows of one group may live on several ranks: local (partial) aggregation, the partial
    groups repartitioned by the group key, then the merge aggregation. Afterwards every group is on exactly one rank."""

    _exchanges = DistributedHashJoinExec._exchanges

    def __init__(self, schema, input: PhysicalPlan, group_exprs, aggregate_exprs):
        from .plan import HashAggregate
        self._schema, self.input = schema, input
        self.group_exprs, self.aggregate_exprs = list(group_exprs), list(aggregate_exprs)
        self.partial = HashAggregate(schema, input, group_exprs, aggregate_exprs)

    def schema(self):
        return self._schema

    def children(self):
        return [self.input]

    def execute_device(self) -> DeviceTable:
        from .expr import Column
        from .plan import HashAggregate
        from .plan import _retrying, exchange_cache
        world = _exchange_world(_dist())
        if not world:
            return _engine().execute(self.partial)
        ng = len(self.group_exprs)
        pschema = self._schema if self._schema is not None else None
        keys = [Column(f"g{k}", k) for k in range(ng)]
        if pschema is None:
            raise _ffi.InternalError(_ffi.QHIP_INVALID_ARGUMENT, "DistributedHashAggregate needs its output schema (the partials travel)")
        E = _engine()
        ctx = E.context()

        def once():
            mine = exchange_cache(ctx).get(id(self))
            if mine is None:
                part = E.execute(self.partial)
                mine = None
                if _fast_exchange():
                    try:
                        mine = shuffle_tables([(part, keys, None, None, 0)])[0]
                    except _ffi.UnsupportedError:
                        mine = None   # (string group keys, NULLs among the partials: the generic path)
                if mine is None:
                    mine = exchange_device_tables(E.partition(part, keys, world), pschema)
                exchange_cache(ctx)[id(self)] = mine
            return E.aggregate(pschema, mine, keys, merge_aggregate_exprs(self.aggregate_exprs, ng))
        return _retrying(ctx, once)

