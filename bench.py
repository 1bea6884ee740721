#!/usr/bin/env python3
"""bench.py — benchmark of the qurious-hip backend on BASELINE.json's metric: rows/s on TPC-H Q1 scan+agg and Q3 hash-join
at SF10, HBM-resident synthetic Arrow tables (contract: the task prompt / DESIGN.md §6).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A STEP is one pass of the hot path over the metric's two configurations: TPC-H Q1's aggregate list over SF10's
59 986 052 lineitem rows (BASELINE configs[2]) followed by TPC-H Q3 at SF10 (configs[3]: customer |><| orders |><|
lineitem -> GROUP BY). `value` = lineitem rows scanned by the two queries / wall time of K such steps, i.e. the job's
whole-pipeline throughput including plan lowering, launches and the host round trips that size intermediate results.
SF10 is fixed, so N > 1 is STRONG scaling: every rank holds a contiguous 1/N slice of every table; Q1 aggregates its
slice and the partial groups are merged (one small all-gather), Q3's joins go through the exchange operators of
qurious_amd/exchange.py over RCCL (--strategy broadcast: all-gather the small build sides; repartition: all-to-all both
sides by key hash; both are measured at N > 1, `value` uses --strategy).

Rank 0 prints ONE JSON line. Besides the contract's fields it carries
  roofline      the step's dominant kernel (Q1's fused filter + aggregate): bytes the kernel actually reads per launch /
                its mean device time (HIP events on the library's stream) against the 8 TB/s HBM peak; `traffic` = HBM
                bytes per launch from this round's PMC profile (profiles/r03_pmc_summary.json) when that profile was taken
                on the same kernel, row count and bytes per row — otherwise null with the reason
  cpu_baseline  the CPU oracle (oracle/qoracle.c, a faithful-cost restatement of the reference executor, 1 thread because
                the reference is single-threaded) on a bounded sample of the same two queries: 1 warm-up + median of 5
  records       one record per configuration — q1_sf10 (configs[2]), q3_sf10 (configs[3]); at N = 1 also q1_mini
                (configs[1], 100 M rows) and filter_lineitem (the standalone Filter operator) — each with its own value,
                per-kernel roofline (bytes actually read) and cpu_baseline
  exchange      (N > 1) bytes each rank sent, seconds inside the exchanges, GB/s against 7 x 153 GB/s of xGMI

Other modes: --workload q1_mini | q1_full | q3 | filter run ONE configuration as the step (profiling, parameter sweeps);
--workload q3 --sf 100 --skew 1.1 --slice R/N generates only rank R's 1/N slice of SF100 (BASELINE configs[4] on one GPU).
"""
import argparse
import json
import os
import statistics
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import qurious_amd as q  # noqa: E402  (before torch: binds the system ROCm runtime first)
from qurious_amd import plan as qplan  # noqa: E402
from qurious_amd import queries, synth  # noqa: E402

METRIC = "rows/s on TPC-H Q1 scan+agg and Q3 hash-join, SF10, 1/2/4/8 MI355X"
SURVEY_BYTES_PER_ROW = {"q1_mini": 25, "q1_full": 78}   # Date32 4 + Utf8 (4+1) [x2 for Q1] + Decimal128 16 [x4]
Q3_SURVEY_BYTES = {"customer": 21, "orders": 28, "lineitem": 44}
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
XGMI_PEAK_GBS = 7 * 153.0                               # 7 links x ~153 GB/s per GPU
SF10_LINEITEM_ROWS = 59_986_052
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_summary.json") for r in (4, 3)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r04_pmc_summary.json"))   # this round's PMC passes (tools/pmc_passes.sh); the last round's until they exist
SETTLE_STEPS = 3   # untimed executions in front of the W warm-up steps: a plan reaches its steady state on its third execution

USE_DIST = False   # set by main(): a torch.distributed process group (RCCL) is up


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def result_key(batches):
    rows = []
    for b in batches:
        rows.extend(zip(*[c.to_pylist() for c in b.columns]))
    return sorted(rows)


def lineitem_table(first_row: int, n_rows: int, batch_rows: int) -> q.MemoryTable:
    starts = list(range(0, n_rows, batch_rows))
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        batches = list(ex.map(lambda s: synth.lineitem_batch(first_row + s, min(batch_rows, n_rows - s)), starts))
    return q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, batches)


def q3_memory_tables(sf, skew, rank, world):
    c, o, l = synth.q3_tables_skewed(sf, skew, rank, world) if skew > 0 else synth.q3_tables(sf, rank, world)
    return (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))


# ---------------------------------------------------------------- timing helpers
class Clock:
    """K timed steps between two barrier + synchronize pairs; MAX over ranks."""

    def __init__(self, ctx, torch, dist):
        self.ctx, self.torch, self.dist = ctx, torch, dist

    def barrier(self):
        if USE_DIST:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.ctx.synchronize()

    def run(self, step, steps, warmup):
        # (a hash join runs with its size left on the device only from the execution after the one that remembered it, and its
        # consumer from the execution after that: three untimed executions in front of the W warm-up steps settle that,
        # whatever W is — first_execution_ms in the records says what the very first one costs)
        for _ in range(SETTLE_STEPS + warmup):
            step()
        self.barrier()
        w0 = self.ctx.sync_count()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.timed_waits = self.ctx.sync_count() - w0   # host waits inside libqhip during the K timed steps (steady state)
        self.ctx.synchronize()   # results are ordered on libqhip's own stream
        self.barrier()
        elapsed = time.perf_counter() - t0
        if USE_DIST:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def operator_stats(step, passes=5):
    """`passes` instrumented (untimed) executions of `step`: per operator call, in execution order, the mean of its
    qhip_exec_stats timings (HIP events on the library's stream) and its last other fields."""
    runs = []
    ctx = q.get_context()
    ctx.set_timing(True)    # (HIP events around the operators' phases: off in the timed region, they cost stream time)
    try:
        for _ in range(passes):
            qplan.STATS_SINK = []
            try:
                step()
            finally:
                sink, qplan.STATS_SINK = qplan.STATS_SINK, None
            runs.append(sink)
    finally:
        ctx.set_timing(False)
    out = []
    for k, (label, st) in enumerate(runs[-1]):
        same = [r[k][1] for r in runs if len(r) == len(runs[-1])]
        rec = dict(st)
        for f in ("main_kernel_ms", "total_device_ms", "build_ms"):
            rec[f] = sum(s[f] for s in same) / len(same)
        rec["operator"] = label
        out.append(rec)
    return out


def pmc_traffic(workload, kernel, rows, bytes_per_row):
    """HBM bytes per launch of `kernel` from this round's PMC profile — only when the profile was taken on the same
    kernel at the same row count and bytes per row (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes)."""
    if not os.path.exists(PMC_SUMMARY):
        return None, f"no {os.path.relpath(PMC_SUMMARY, ROOT)}"
    try:
        with open(PMC_SUMMARY) as f:
            prof = json.load(f).get(workload, {})
    except Exception as e:
        return None, f"unreadable profile summary: {e}"
    seen = []
    for k in prof.get("kernels", []):
        if k.get("kernel") != kernel:
            continue
        if int(k.get("rows", -1)) == int(rows) and abs(float(k.get("kernel_bytes_per_row", -1)) - float(bytes_per_row)) < 1e-6:
            return k.get("hbm_bytes_per_launch"), f"{os.path.relpath(PMC_SUMMARY, ROOT)} ({prof.get('collected', '?')})"
        seen.append(f"rows={k.get('rows')} bytes/row={k.get('kernel_bytes_per_row')}")
    if seen:
        return None, f"profile has {kernel} at {'; '.join(seen)} — this run: rows={rows} bytes/row={bytes_per_row}"
    return None, f"{kernel} not in the profile summary of {workload}"


def roofline(kernel, kernel_ms, bytes_per_launch, traffic=None, traffic_source=None, algorithmic_bytes_per_launch=None, **extra):
    """`achieved` / `frac` = the bytes the kernel READS per launch (qhip_exec_stats.bytes_per_row_read x rows: what it must pull
    from HBM in the layout that is resident and streamed, checked against the PMC `traffic`) / its mean duration. Where the
    kernel streams narrow copies of Decimal128 / Int64 columns this is SURVEY §8(d)'s rule for a narrower device
    layout: the algorithmic bytes are recomputed from the bytes actually resident — Arrow-layout bytes are never divided by a
    narrowed kernel's time. The Arrow layout's figure is reported beside it for reference only (`arrow_layout_bytes_per_launch`)."""
    per_s = 1.0 / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": bytes_per_launch * per_s, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_launch * per_s / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel, "kernel_ms": kernel_ms,
         "bytes_read_per_launch": bytes_per_launch}
    if algorithmic_bytes_per_launch is not None:
        r["arrow_layout_bytes_per_launch"] = algorithmic_bytes_per_launch
    r.update(extra)
    return r


def cold_and_export_ms(ctx, make_plan, runs=3):
    """(first_execution_ms, execute_with_export_ms): the first execution of a NEW plan object after the context forgot what it
    had learnt about earlier plans (qhip_ctx_forget_plans: lowered plans, join sizes, group counts — compiled kernels stay
    loaded, table-column statistics stay), through to the synchronised device result; and a repeated `execute()` of a warm
    plan through to host Arrow batches (the reference's contract: PhysicalPlan::execute -> Vec<RecordBatch>). Medians of `runs`."""
    cold, export = [], []
    for _ in range(runs):
        ctx.synchronize()
        ctx.forget_plans()
        plan = make_plan()
        t0 = time.perf_counter()
        plan.execute_device()
        ctx.synchronize()
        cold.append((time.perf_counter() - t0) * 1e3)
    plan = make_plan()
    for _ in range(SETTLE_STEPS):
        plan.execute()
    for _ in range(runs):
        t0 = time.perf_counter()
        plan.execute()
        export.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(cold), statistics.median(export)


def cold_table_ms(ctx, make_plan, tables, runs=2):
    """What a query over a FRESHLY UPLOADED table pays: the first three executions of a new plan after the context forgot its
    plans AND the tables forgot their column statistics and narrow copies (qhip_table_forget_statistics) — execution 1 collects
    the statistics (one fused pass per column that also writes the speculative narrow copy), 2 and 3 settle the plan. Returns
    {cold_first_query_ms, cold_queries_ms: [q1, q2, q3], table_prepare_ms: q1 minus the steady execution, aux_bytes}."""
    seqs = []
    for _ in range(runs):
        ctx.synchronize()
        ctx.forget_plans()
        for t in tables:
            t.device_table().forget_statistics()
        plan = make_plan()
        seq = []
        for _ in range(3):
            t0 = time.perf_counter()
            plan.execute_device()
            ctx.synchronize()
            seq.append((time.perf_counter() - t0) * 1e3)
        seqs.append(seq)
    best = min(seqs, key=lambda q: q[0])
    for _ in range(2):
        plan.execute_device()
    ctx.synchronize()
    t0 = time.perf_counter()
    plan.execute_device()
    ctx.synchronize()
    steady = (time.perf_counter() - t0) * 1e3
    return {"cold_first_query_ms": best[0], "cold_queries_ms": best, "steady_query_ms_with_sync": steady,
            "table_prepare_ms": max(0.0, best[0] - steady),
            "aux_bytes": sum(t.device_table().aux_bytes for t in tables),
            "note": "cold = plans forgotten AND the tables' column statistics / narrow copies dropped (a freshly uploaded table; compiled "
                    "kernels stay loaded); aux_bytes = HBM the narrow copies occupy beside the Arrow-layout buffers"}


def median_runs(fn, runs=5):
    """1 warm-up + `runs` timed calls of fn() -> seconds; returns (median seconds, all timed seconds, last result)."""
    res = fn()
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        res = fn()
        times.append(time.perf_counter() - t0)
    return statistics.median(times), times, res


# ---------------------------------------------------------------- CPU baselines (rank 0, N = 1 only)
def cpu_scan_aggregate(workload, table, batch_rows, sample_rows):
    """The oracle's qo_scan_filter_aggregate (per-batch literal broadcast + cast + compare, compaction of all 7 columns,
    concat, SipHash per row, hash -> group map, per-group take + reduce) over the first `sample_rows` rows."""
    from oracle import qoracle
    sample = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, table.data[: max(1, sample_rows // batch_rows)])
    cplan = getattr(queries, workload)(sample)
    rows = sum(b.num_rows for b in sample.data)
    log(f"cpu baseline {workload}: oracle over {rows} rows, 1 warm-up + 5 runs ...")
    med, times, res = median_runs(lambda: qoracle.scan_filter_aggregate_timed(cplan))
    assert result_key([res[0]]) == result_key(cplan.execute()), f"HIP {workload} differs from the CPU oracle on the baseline sample"
    log(f"cpu baseline {workload}: {rows / med / 1e6:.2f} M rows/s (median {med:.2f} s)")
    return {"value": rows / med, "unit": "rows/s", "cores": 1, "kind": "port", "runs": len(times), "seconds_median": med,
            "seconds_all": [round(t, 3) for t in times],
            "sample": f"{rows} rows ({len(sample.data)} batches of {batch_rows}) of the same workload through oracle/qoracle.c "
                      f"qo_scan_filter_aggregate, 1 of {os.cpu_count()} host cores (the reference executor is single-threaded), "
                      "1 warm-up + median of 5; HIP result on the sample equals the oracle's bit for bit"}


def cpu_q3(sf):
    from oracle import qoracle
    tabs = q3_memory_tables(sf, 0.0, 0, 1)
    splan = queries.q3(*tabs)
    nl = sum(b.num_rows for b in tabs[2].data)
    log(f"cpu baseline q3: oracle at SF{sf} ({nl} lineitem rows), 1 warm-up + 5 runs ...")
    med, times, want = median_runs(lambda: qoracle.execute(splan))
    assert result_key(want) == result_key(splan.execute()), "HIP Q3 result differs from the CPU oracle on the baseline sample"
    log(f"cpu baseline q3: {nl / med / 1e6:.2f} M lineitem rows/s (median {med:.2f} s)")
    return {"value": nl / med, "unit": "rows/s", "cores": 1, "kind": "port", "runs": len(times), "seconds_median": med,
            "seconds_all": [round(t, 3) for t in times],
            "sample": f"Q3 at SF{sf} ({nl} lineitem rows) through oracle/qoracle.py + qoracle.c (JoinHashMap chains, SipHash, per-group "
                      f"take + reduce), 1 of {os.cpu_count()} host cores, 1 warm-up + median of 5; HIP result on the sample equals the oracle's"}


# ---------------------------------------------------------------- configurations
class Q1:
    """configs[1] (q1_mini, 100 M rows) / configs[2] (q1_full = TPC-H Q1's aggregate list, SF10 rows): the fused filter +
    hash-aggregate kernel over this rank's slice of the table."""

    def __init__(self, args, ctx, workload, rows_total, rank, world):
        self.workload, self.rows_total, self.world, self.ctx = workload, rows_total, world, ctx
        lo = rows_total * rank // world
        self.rows = rows_total * (rank + 1) // world - lo
        t0 = time.time()
        self.table = lineitem_table(lo, self.rows, args.batch_rows)
        self.t_gen = time.time() - t0
        t0 = time.time()
        dev = self.table.device_table()
        self.t_upload = time.time() - t0
        self.resident = sum(dev.column_bytes(c) for c in range(dev.num_columns))
        # the same upload once more into a second table (dropped at once): the first one pays ~40 ms of hipMalloc page mapping
        # for the pool's first 4.7 GB on top of the transfer, this one shows the transfer (DESIGN §6 "Upload")
        t0 = time.time()
        again = q.DeviceTable.from_batches(ctx, self.table.schema(), self.table.data)
        ctx.synchronize()
        self.t_upload_warm = time.time() - t0
        del again
        # N > 1: every rank aggregates its slice into partial groups (AVG planned as SUM and COUNT) that are merged after one
        # small all-gather; N = 1: the query as the reference's planner builds it
        self.plan = queries.q1_partial(self.table) if USE_DIST and workload == "q1_full" else getattr(queries, workload)(self.table)
        self.batch_rows = args.batch_rows
        log(f"{workload}: {self.rows} rows generated in {self.t_gen:.1f}s, uploaded in {self.t_upload:.1f}s")

    def step(self):
        out = self.plan.execute_device()
        if USE_DIST:
            self.merged = merge_partial_groups(self.plan, out)
        return out

    def record(self, args, clock, with_cpu):
        elapsed = clock.run(self.step, args.steps, args.warmup)
        ops = operator_stats(self.step)
        st = ops[-1]
        bpr = min(SURVEY_BYTES_PER_ROW[self.workload], st.get("bytes_per_row_read") or SURVEY_BYTES_PER_ROW[self.workload])
        kbytes = self.rows * bpr
        traffic, src = pmc_traffic(self.workload, st["main_kernel_name"], self.rows, bpr)
        cfg = "configs[1]" if self.workload == "q1_mini" else "configs[2]"
        what = ("filter + GROUP BY l_returnflag, SUM(l_quantity)" if self.workload == "q1_mini" else
                "TPC-H Q1: 2 keys, 4 SUM + 3 AVG + COUNT (q1.slt:5-12)")
        rec = {"workload": f"{cfg} {self.workload}: {what} over {self.rows_total} synthetic lineitem rows "
                           f"({args.batch_rows}-row Arrow batches, HBM-resident, {self.world} rank(s))",
               "value": self.rows_total * args.steps / elapsed, "unit": "rows/s", "steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3,
               "rows": self.rows_total, "groups": st["groups"], "resident_bytes_per_gpu": self.resident,
               "roofline": roofline(st["main_kernel_name"], st["main_kernel_ms"], kbytes, traffic, src,
                                    algorithmic_bytes_per_launch=self.rows * SURVEY_BYTES_PER_ROW[self.workload],
                                    arrow_layout_bytes_per_row=SURVEY_BYTES_PER_ROW[self.workload], kernel_bytes_per_row=bpr,
                                    rows_per_launch=self.rows,
                                    layout=("narrow copies of the Decimal128 columns (4 / 8 bytes per value)" if bpr < 0.75 * SURVEY_BYTES_PER_ROW[self.workload]
                                            else "Arrow layout")),
               "cpu_baseline": None}
        if not USE_DIST:
            rec["first_execution_ms"], rec["execute_with_export_ms"] = cold_and_export_ms(self.ctx, lambda: getattr(queries, self.workload)(self.table))
            rec["cold_table"] = cold_table_ms(self.ctx, lambda: getattr(queries, self.workload)(self.table), [self.table])
        if with_cpu:
            rec["cpu_baseline"] = cpu_scan_aggregate(self.workload, self.table, self.batch_rows, args.cpu_sample_rows)
        return rec, elapsed


def merge_partial_groups(plan, out):
    """N > 1, Q1: all-gather the ranks' partial groups (<= 16 groups x 8 int64 words) and merge them on the host: SUM of
    sums, SUM of counts (the Decimal128 sums as two int64 halves). Returns {key: [sums..., count]}."""
    import torch
    import torch.distributed as dist
    rows = result_key(out.to_batches())
    nk = len(plan.group_exprs)
    words = 2 * (len(rows[0]) - nk) if rows else 0
    buf = torch.zeros((16, 2 + max(words, 14)), dtype=torch.int64)
    for g, r in enumerate(rows[:16]):
        key = "|".join(str(v) for v in r[:nk]).encode()[:8]
        buf[g, 0] = int.from_bytes(key.ljust(8, b"\0"), "little", signed=True)
        buf[g, 1] = 1
        for k, v in enumerate(r[nk:]):
            u = int(v.scaleb(-v.as_tuple().exponent)) if hasattr(v, "as_tuple") else int(v)
            u &= (1 << 128) - 1
            lo, hi = u & ((1 << 64) - 1), u >> 64
            buf[g, 2 + 2 * k] = lo - (1 << 64) if lo >= (1 << 63) else lo
            buf[g, 3 + 2 * k] = hi - (1 << 64) if hi >= (1 << 63) else hi
    dev = buf.cuda()
    parts = [torch.empty_like(dev) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, dev)
    merged = {}
    for p in parts:
        for row in p.cpu().tolist():
            if not row[1]:
                continue
            acc = merged.setdefault(row[0], [0] * (len(row) // 2 - 1))
            for k in range(len(acc)):
                acc[k] = (acc[k] + ((row[2 + 2 * k] & ((1 << 64) - 1)) | ((row[3 + 2 * k] & ((1 << 64) - 1)) << 64))) & ((1 << 128) - 1)
    return merged


class Q3:
    """configs[3] / configs[4]: TPC-H Q3 (customer |><| orders |><| lineitem -> GROUP BY) at --sf, every table sliced over
    the ranks; with N > 1 the joins go through the exchange operators (qurious_amd/exchange.py)."""

    def __init__(self, args, ctx, rank, world, strategy, slice_of=None):
        from qurious_amd import exchange
        self.exchange, self.ctx, self.world, self.strategy, self.sf, self.skew = exchange, ctx, world, strategy, args.sf, args.skew
        t0 = time.time()
        r, w = slice_of if slice_of else (rank, world)
        self.tabs = q3_memory_tables(args.sf, args.skew, r, w)
        self.rows = [sum(b.num_rows for b in t.data) for t in self.tabs]
        for t in self.tabs:
            t.device_table()
        log(f"q3: SF{args.sf} slice {r}/{w} {self.rows} rows generated + uploaded in {time.time() - t0:.1f}s")
        self.slice_of = slice_of
        self.plans = {}

    def plan(self, strategy):
        if strategy not in self.plans:
            ex = self.exchange
            if USE_DIST and strategy == "broadcast":
                # the small build sides are all-gathered, the big probe sides stay where they are, partial groups are merged
                p = queries.q3(*self.tabs, join_cls=ex.BroadcastHashJoinExec, agg_cls=ex.DistributedHashAggregate)
            elif USE_DIST and strategy == "range":
                # join 1 broadcasts the customer keys and leaves the orders in place; join 2 leaves BOTH big tables in place and sends
                # a build row only to the ranks whose lineitem keys can reach it (RangeBroadcastHashJoinExec, QHIP_EXCHANGE_RANGE,
                # DESIGN §7): tables sliced in key order exchange the orders at their slice borders; partial groups are merged
                p = queries.q3(*self.tabs, join_cls=ex.BroadcastHashJoinExec, join2_cls=ex.RangeBroadcastHashJoinExec, agg_cls=ex.DistributedHashAggregate)
            else:
                p = queries.q3(*self.tabs, join_cls=ex.DistributedHashJoinExec if USE_DIST else None)
            if USE_DIST:
                ex.prune_exchange_columns(p)   # the exchanges move only the columns the plan above them reads
            self.plans[strategy] = p
        return self.plans[strategy]

    def step(self, strategy=None):
        strategy = strategy or self.strategy
        if strategy == "range":
            os.environ["QHIP_EXCHANGE_RANGE"] = "1"
        try:
            self.out = self.plan(strategy).execute_device()
        finally:
            os.environ.pop("QHIP_EXCHANGE_RANGE", None)
        return self.out

    def totals(self, torch, dist):
        vals = self.rows + [self.out.num_rows]
        if USE_DIST:
            t = torch.tensor(vals, dtype=torch.int64, device="cuda")
            dist.all_reduce(t)
            vals = t.tolist()
        return vals

    def record(self, args, clock, torch, dist, with_cpu, strategy=None, extras=True):
        strategy = strategy or self.strategy
        ex = self.exchange
        if USE_DIST:
            ex.exchange_stats(reset=True)
        elapsed = clock.run(lambda: self.step(strategy), args.steps, args.warmup)
        lib_waits = clock.timed_waits            # over the K timed steps (the settling executions wait more: sizes not learnt yet)
        xg = ex.exchange_stats(reset=True) if USE_DIST else None
        tot = self.totals(torch, dist)
        ops = operator_stats(lambda: self.step(strategy), passes=3)
        joins = [o for o in ops if o["operator"] == "hash_join"]
        aggs = [o for o in ops if o["operator"] == "aggregate"]
        kernels = []
        for k, j in enumerate(joins):
            pb = j["rows_in"] * j["bytes_per_row_read"]
            pname = j["main_kernel_name"]
            dense = pname.startswith("qk_join_probe_dense")
            t, src = pmc_traffic("q3", pname, j["rows_in"], j["bytes_per_row_read"])
            if t is not None:
                src += ("; FETCH_SIZE x 2 is calibrated for wide streaming reads (MI355X_MICROARCH.md): the random 4 / 8 / 16-byte lookups of "
                        "a probe make its traffic an UPPER bound")
            kernels.append(dict(roofline(pname, j["main_kernel_ms"], pb, t, src, kernel_bytes_per_row=j["bytes_per_row_read"],
                                         rows_per_launch=j["rows_in"]), operator=f"join {k + 1} probe", pairs=j["groups"],
                                table_layout="dense: exact bitmap over the key range + row_of[key - min]" if dense else "hashed: open addressing + blocked filter"))
            if dense and os.environ.get("QHIP_JOIN_DENSE_BYTEMAP", "0") not in ("0", ""):
                # key columns read + one stamp byte and one row_of entry written per build row + the byte map read and the bitmap written
                bb = j["build_rows"] * (j["build_bytes_per_row"] + 5.0) + j["table_capacity"] * (1.0 + 1.0 / 8.0)
                bname = "qk_join_dense_build + k_bytes_to_bits"
            elif dense:   # key columns read + bitmap cleared and set + one row_of entry written per build row
                bb = j["build_rows"] * (j["build_bytes_per_row"] + 4.0) + j["table_capacity"] / 8.0 * 2
                bname = "memset + qk_join_dense_build"
            else:       # key columns read + table and filter written
                bb = j["build_rows"] * j["build_bytes_per_row"] + j["table_capacity"] * 17.0
                bname = "qk_join_scatter + k_join_region_build"
            kernels.append(dict(roofline(bname, j["build_ms"], bb, None, "not profiled per launch",
                                         rows_per_launch=j["build_rows"], table_slots=j["table_capacity"]), operator=f"join {k + 1} build"))
        for a in aggs:
            kernels.append(dict(roofline(a["main_kernel_name"], a["main_kernel_ms"], a["rows_in"] * a["bytes_per_row_read"], None,
                                         "not profiled per launch", kernel_bytes_per_row=a["bytes_per_row_read"], rows_per_launch=a["rows_in"]),
                                operator="aggregate", groups=a["groups"]))
        dominant = max(kernels, key=lambda k: k["kernel_ms"]) if kernels else None
        survey = tot[0] * Q3_SURVEY_BYTES["customer"] + tot[1] * Q3_SURVEY_BYTES["orders"] + tot[2] * Q3_SURVEY_BYTES["lineitem"]
        cfg = "configs[4]" if self.skew > 0 else "configs[3]"
        par = ("one GPU" if not USE_DIST else f"broadcast build sides, local probes, merged partial groups x{self.world}" if strategy == "broadcast"
               else f"both join sides repartitioned by key hash (all-to-all) x{self.world}")
        rec = {"workload": f"{cfg} q3: TPC-H Q3 (q3.slt:1-24) SF{self.sf} customer|><|orders|><|lineitem + GROUP BY, HBM-resident"
                           + (f", join keys re-drawn from Zipf(s={self.skew})" if self.skew > 0 else "")
                           + (f", slice {self.slice_of[0]}/{self.slice_of[1]} of every table on one GPU" if self.slice_of else "") + f"; {par}",
               "value": tot[2] * args.steps / elapsed, "unit": "lineitem rows/s", "steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3,
               "rows": {"customer": tot[0], "orders": tot[1], "lineitem": tot[2]}, "groups": tot[3], "strategy": strategy if USE_DIST else None,
               "roofline": dominant, "kernels": kernels,
               "survey_algorithmic_bytes": survey, "survey_algorithmic_GBps_of_wall_time": survey / (elapsed / args.steps) / 1e9 / self.world,
               "device_ms_per_query": {"join": [j["total_device_ms"] for j in joins], "aggregate_kernel": [a["main_kernel_ms"] for a in aggs]},
               "host_waits_per_query": lib_waits / args.steps,
               "cpu_baseline": None}
        if not USE_DIST and not self.slice_of and extras:
            rec["first_execution_ms"], rec["execute_with_export_ms"] = cold_and_export_ms(self.ctx, lambda: queries.q3(*self.tabs))
            rec["cold_table"] = cold_table_ms(self.ctx, lambda: queries.q3(*self.tabs), list(self.tabs))
        if xg:
            nq = args.steps + args.warmup + SETTLE_STEPS   # queries since the counters were reset
            secs = max(xg.get("seconds", 0.0), 1e-12)
            rec["exchange"] = {"bytes_sent_per_query": xg.get("bytes_sent", 0) / nq, "bytes_received_per_query": xg.get("bytes_received", 0) / nq,
                               "seconds_per_query": xg.get("seconds", 0.0) / nq, "exchanges_per_query": xg.get("exchanges", 0) / nq,
                               "send_GBps": xg.get("bytes_sent", 0) / secs / 1e9, "xgmi_peak_GBps": XGMI_PEAK_GBS,
                               "frac_of_xgmi": xg.get("bytes_sent", 0) / secs / 1e9 / XGMI_PEAK_GBS,
                               "transport": ex.transport(), "rccl_version_seen_by_libqhip": xg.get("rccl_version"),
                               # host waits of a query: stream synchronisations inside libqhip (its own transport's included)
                               # plus, with the torch transport, torch's header read-backs / synchronisations
                               "host_waits_per_query": lib_waits / args.steps + (xg.get("transport_waits", 0) / nq if ex.transport() == "torch" else 0),
                               "transport_waits_per_query": xg.get("transport_waits", 0) / nq,
                               "heavy_key_rounds_per_query": xg.get("heavy_key_rounds", 0) / nq, "note": "rank 0 only"}
        if with_cpu:
            rec["cpu_baseline"] = cpu_q3(min(self.sf, args.cpu_q3_sf))
        return rec, elapsed

    def table_stats(self):
        """BASELINE configs[4]: LDS hash-table occupancy of the final aggregate (one extra, untimed, instrumented execution)"""
        os.environ["QHIP_AGG_STATS"] = "1"
        try:
            self.step()
            st = self.ctx.last_stats()
        finally:
            os.environ.pop("QHIP_AGG_STATS", None)
        return {"lds_table_slots_per_workgroup": st["lds_table_slots"], "lds_occupancy": st["lds_occupancy"], "lds_spilled": bool(st["lds_spilled"]),
                "hbm_table_slots": st["table_capacity"], "hbm_table_load": st["hbm_table_load"], "groups": st["groups"], "workgroups": st["workgroups"]}


class FilterBench:
    """The standalone Filter operator (physical/plan/filter.rs:28-44): predicate -> ballot mask -> scan -> selection vector ->
    every column gathered, over the 7-column lineitem table; a ~98 % and a ~1 % selective predicate."""

    def __init__(self, args, ctx, table):
        import pyarrow as pa
        from qurious_amd import BinaryExpr, CastExpr, Column, Literal, Operator, ScalarValue
        self.ctx, self.table, self.rows = ctx, table, sum(b.num_rows for b in table.data)
        day = lambda s: CastExpr(Literal(ScalarValue.Utf8(s)), pa.date32())   # noqa: E731
        self.plans = {"keeps ~98 %": q.Filter(q.Scan(synth.LINEITEM_SCHEMA, table, None, None), BinaryExpr(Column("l_shipdate", 0), Operator.LtEq, day("1998-09-02"))),
                      "keeps ~1 %": q.Filter(q.Scan(synth.LINEITEM_SCHEMA, table, None, None), BinaryExpr(Column("l_shipdate", 0), Operator.Lt, day("1992-01-27")))}

    def record(self, args, clock):
        out = {}
        for name, plan in self.plans.items():
            elapsed = clock.run(lambda: plan.execute_device(), max(3, args.steps // 2), 2)
            steps = max(3, args.steps // 2)
            qplan.STATS_SINK = None
            self.ctx.set_timing(True)    # (one instrumented, untimed execution: HIP events around the operator)
            try:
                res = plan.execute_device()
                st = self.ctx.last_stats()
            finally:
                self.ctx.set_timing(False)
            kept = res.num_rows
            # bytes the operator moves: the predicate column once, then per kept row every column's value read and written + the
            # 4-byte selection index read once per column
            dev = self.table.device_table()
            per_row_all = sum(dev.column_bytes(c) for c in range(dev.num_columns)) / max(1, self.rows)
            moved = self.rows * 4 + self.rows / 8 * 2 + kept * 4 + kept * (2 * per_row_all + 4 * dev.num_columns)
            out[name] = {"rows_in": self.rows, "rows_out": kept, "ms_per_call": elapsed / steps * 1e3, "rows_per_s": self.rows * steps / elapsed,
                         "device_ms": st["total_device_ms"],
                         "roofline": roofline("qk_pred_mask + scan + k_select_indices + k_gather_*", st["total_device_ms"], moved, None,
                                              "not profiled per launch", bytes_are="read + written")}
        return {"workload": f"standalone Filter (filter.rs:28-44) over {self.rows} lineitem rows x 7 columns, HBM-resident", "cases": out}


class PartitionBench:
    """The exchange's first step on one rank (qhip_partition_filtered): Q3's lineitem side — scan filter
    l_shipdate > 1995-03-15, key l_orderkey, the three columns the plan above the exchange reads — split into 8 parts, over the
    whole SF10 table (what one rank of one holds) and over rank 0's 1/8 slice (what one rank of eight holds)."""

    def __init__(self, ctx, tabs_full=None):
        self.ctx = ctx
        self.tabs = {}
        if tabs_full is not None:
            self.tabs["sf10"] = tabs_full
        else:
            self.tabs["sf10"] = q3_memory_tables(10.0, 0.0, 0, 1)
        self.tabs["sf10_slice_1_of_8"] = q3_memory_tables(10.0, 0.0, 0, 8)

    def record(self, args, clock, n_parts=8):
        from qurious_amd import exchange
        out = {}
        for name, tabs in self.tabs.items():
            plan = queries.q3(*tabs)
            j2 = plan.input
            scan, key = j2.right, j2.on[0][1]
            schema = scan.schema()
            need = exchange.referenced_columns(list(plan.group_exprs) + [a.expression() for a in plan.aggregate_exprs])
            nl = len(j2.left.schema())
            keep = [(nl + c) in need or c == key.index for c in range(len(schema))]
            dev = tabs[2].device_table()
            rows = dev.num_rows
            run = lambda: exchange.partition_filtered(dev, [key], n_parts, predicate=scan.filter, keep=keep)   # noqa: E731
            steps = max(5, args.steps)
            elapsed = clock.run(run, steps, 2)
            self.ctx.set_timing(True)
            try:
                p1, p2, tot = [], [], []
                for _ in range(5):
                    parts = run()
                    st = self.ctx.last_stats()
                    p1.append(st["build_ms"]); p2.append(st["main_kernel_ms"]); tot.append(st["total_device_ms"])
            finally:
                self.ctx.set_timing(False)
            kept = sum(p.num_rows for p in parts)
            widths = [pa_width(f.type) for f in schema]
            pred_cols = exchange.referenced_columns([scan.filter, key])
            w_kept = sum(w for w, k in zip(widths, keep) if k)
            w_read_once = sum(w for c, w in enumerate(widths) if keep[c] or c in pred_cols)
            algorithmic = rows * w_read_once + kept * w_kept           # every referenced column read once + the kept rows written once
            moved = rows * (st["build_bytes_per_row"] + 1) + rows * (1 + w_kept) + kept * w_kept   # incl. the part byte written and re-read, key read twice
            k1, k2 = statistics.median(p1), statistics.median(p2)
            out[name] = {"rows_in": rows, "rows_out": kept, "parts": n_parts, "part_rows": [p.num_rows for p in parts],
                         "columns_moved": [f.name for f, k in zip(schema, keep) if k],
                         "ms_per_call": elapsed / steps * 1e3, "rows_per_s": rows * steps / elapsed,
                         "pass1_ms": k1, "pass2_ms": k2, "device_ms": statistics.median(tot), "host_waits_per_call": 1,
                         "device_ms_is": "first launch .. last launch of a call: the two kernels + the one-workgroup scan of the (part, unit) counters "
                                         "+ the host wait that sizes the parts (~30 us)",
                         "roofline": roofline("qk_part_ids_wide + " + st["main_kernel_name"], k1 + k2, algorithmic, None, "see pass1 / pass2",
                                              bytes_are="every referenced column read once + the kept rows' kept columns written once",
                                              bytes_moved_per_launch=moved, frac_on_bytes_moved=moved / ((k1 + k2) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              pass1=self._pass(name, "qk_part_ids_wide" if rows >= 2048 else "qk_part_ids", k1, rows, st["build_bytes_per_row"] + 1),
                                              pass2=self._pass(name, st["main_kernel_name"], k2, rows, (rows * (1 + w_kept) + kept * w_kept) / rows))}
        return {"workload": f"exchange step 1 (SURVEY §8e): Q3's lineitem side, scan filter + partition by mix64(l_orderkey) into {n_parts} parts, "
                            "HBM-resident; no reference counterpart (single process)", "cases": out}


def _partition_pass(self, case, kernel, ms, rows, bytes_per_row):
    """one pass of the partition step: bytes moved by construction per input row (reads + writes) over its mean duration, with
    the PMC traffic of the same kernel at the same rows when this round's profile has it"""
    traffic, source = pmc_traffic("partition_" + case, kernel, rows, round(bytes_per_row, 6))
    return roofline(kernel, ms, rows * bytes_per_row, traffic, source, rows_per_launch=rows, kernel_bytes_per_row=round(bytes_per_row, 6),
                    bytes_are="column bytes read + part byte written (pass 1) / part byte + kept columns read, kept rows written (pass 2)")


PartitionBench._pass = _partition_pass


def pa_width(t):
    import pyarrow as pa
    return 16 if pa.types.is_decimal128(t) else t.bit_width // 8


def cpp_host_step(args):
    """The SAME step (Q1 at SF10 + Q3 at SF10) driven by the compiled host — tools/bench_host: the C++ mirror of the reference's
    operator API (include/qhip_plan.hpp) over the C ABI — in a child process with its own context and tables: what the host
    language costs the metric (the timed region of this file goes through the Python mirror)."""
    exe = os.path.join(ROOT, "tools", "bench_host")
    if not os.path.exists(exe):
        return {"skipped": "tools/bench_host is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    import subprocess
    r = subprocess.run([exe, str(args.steps), str(args.warmup + SETTLE_STEPS)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    rec["note"] = "child process: own context, own copy of the tables; one 60 M-row batch per table instead of 2^20-row batches"
    return rec


# ---------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="metric", choices=["metric", "q1_mini", "q1_full", "q3", "filter", "partition"],
                    help="metric (default): a step = Q1 at SF10 + Q3 at SF10; the others run one configuration as the step")
    ap.add_argument("--rows", type=int, default=0, help="lineitem rows of the q1_* / filter workloads (whole job; default: 100 M for q1_mini, SF10's for q1_full)")
    ap.add_argument("--batch-rows", type=int, default=1 << 20)
    ap.add_argument("--sf", type=float, default=10.0, help="TPC-H scale factor of Q3 (whole job, sliced over the ranks)")
    ap.add_argument("--strategy", default="broadcast", choices=["broadcast", "repartition", "range"],
                    help="Q3 on several GPUs: all-gather the small build sides (default), repartition both sides of every join by key hash, "
                         "or broadcast join 1's build side and send join 2's build rows only where the ranks' probe key ranges can reach them")
    ap.add_argument("--skew", type=float, default=0.0, help="Q3: re-draw the join keys from Zipf(s) (configs[4] uses 1.1); 0 = uniform")
    ap.add_argument("--slice", default="", help="Q3 on ONE GPU over rank R's 1/N slice of every table, as R/N (configs[4]: --sf 100 --skew 1.1 --slice 0/8)")
    ap.add_argument("--cpu-sample-rows", type=int, default=16 << 20, help="rows of a q1_* workload timed through the CPU oracle (per run)")
    ap.add_argument("--cpu-q3-sf", type=float, default=2.0, help="scale factor of the Q3 sample timed through the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the q1_mini and Filter records")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("QHIP_DEVICE", str(local_rank))
    ctx = q.get_context()   # raises without a gfx950 device: there is no CPU fallback
    log(f"context on {ctx.device_name()}")

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    # QHIP_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, collectives, exchange operators) even
    # with ONE rank — the rehearsal of the N > 1 launch that fits a one-GPU box (tests/test_gpu_q3.py runs it)
    global USE_DIST
    USE_DIST = world > 1 or os.environ.get("QHIP_BENCH_FORCE_DIST") == "1"
    rccl = None
    if USE_DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
    clock = Clock(ctx, torch, dist)
    with_cpu = not args.no_cpu_baseline and world == 1 and not USE_DIST
    log("torch imported")
    line = {"metric": METRIC, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "vs_baseline": None, "dtype": "i128", "data": "synthetic", "device": ctx.device_name(),
            "state": f"warm: the timed steps repeat cached plans over resident tables ({SETTLE_STEPS} untimed settling executions + W warm-up "
                     "steps first: learnt join sizes, one host wait per query, remembered group counts); records.*.first_execution_ms = a new "
                     "plan with that knowledge forgotten (code cache warm), records.*.execute_with_export_ms = through execute() to host Arrow batches"}
    if USE_DIST:
        line["distributed"] = {"world_size": dist.get_world_size(), "backend": "nccl (RCCL)", "rccl_version": rccl}

    def finish():
        if rank == 0:
            print(json.dumps(line))
        if USE_DIST:
            dist.destroy_process_group()

    # ---- single-configuration modes (profiling / sweeps): the chosen configuration is the step
    if args.workload in ("q1_mini", "q1_full"):
        rows = args.rows or (100_000_000 if args.workload == "q1_mini" else SF10_LINEITEM_ROWS)
        w = Q1(args, ctx, args.workload, rows, rank, world)
        rec, elapsed = w.record(args, clock, with_cpu)
        line.update(value=rec["value"], ms_per_step=rec["ms_per_step"], scaling="strong", roofline=rec["roofline"], cpu_baseline=rec["cpu_baseline"],
                    config={"workload": rec["workload"], "rows": rows, "batch_rows": args.batch_rows, "groups": rec["groups"],
                            "parallelism": f"row-range slices x{world}, partial groups merged"},
                    setup_s={"generate": w.t_gen, "upload_h2d": w.t_upload, "h2d_GBps": w.resident / max(w.t_upload, 1e-9) / 1e9,
                             "h2d_GBps_warm_allocator": w.resident / max(w.t_upload_warm, 1e-9) / 1e9})
        return finish()
    if args.workload == "filter":
        rows = args.rows or SF10_LINEITEM_ROWS
        w = Q1(args, ctx, "q1_full", rows, rank, world)
        rec = FilterBench(args, ctx, w.table).record(args, clock)
        first = next(iter(rec["cases"].values()))
        line.update(value=first["rows_per_s"], ms_per_step=first["ms_per_call"], scaling="strong", roofline=first["roofline"], cpu_baseline=None,
                    config={"workload": rec["workload"]}, records={"filter_lineitem": rec})
        return finish()
    if args.workload == "partition":
        rec = PartitionBench(ctx).record(args, clock)
        first = rec["cases"]["sf10"]
        line.update(value=first["rows_per_s"], ms_per_step=first["ms_per_call"], scaling="strong", roofline=first["roofline"], cpu_baseline=None,
                    config={"workload": rec["workload"]}, records={"partition": rec})
        return finish()
    if args.workload == "q3":
        slice_of = tuple(int(x) for x in args.slice.split("/")) if args.slice else None
        w = Q3(args, ctx, rank, world, args.strategy, slice_of)
        rec, elapsed = w.record(args, clock, torch, dist, with_cpu)
        line.update(value=rec["value"], ms_per_step=rec["ms_per_step"], scaling="strong", roofline=rec["roofline"], cpu_baseline=rec["cpu_baseline"],
                    config={"workload": rec["workload"], "rows": rec["rows"], "groups": rec["groups"]}, records={"q3": rec})
        if args.skew > 0 or os.environ.get("QHIP_AGG_STATS"):
            line["aggregate_table"] = w.table_stats()
        if "exchange" in rec:
            line["exchange"] = rec["exchange"]
        return finish()

    # ---- the metric: a step = Q1 at SF10 (configs[2]) + Q3 at SF10 (configs[3])
    q1 = Q1(args, ctx, "q1_full", SF10_LINEITEM_ROWS, rank, world)
    args.sf, args.skew = 10.0, 0.0
    q3 = Q3(args, ctx, rank, world, args.strategy)

    def step():
        q1.step()
        q3.step()

    elapsed = clock.run(step, args.steps, args.warmup)
    tot = q3.totals(torch, dist)
    rows_step = SF10_LINEITEM_ROWS + tot[2]
    log(f"timed {args.steps} steps in {elapsed:.3f}s")
    records = {}
    records["q1_sf10"], _ = q1.record(args, clock, with_cpu)
    records["q3_sf10"], _ = q3.record(args, clock, torch, dist, with_cpu)
    if USE_DIST:
        for other in ("broadcast", "repartition", "range"):
            if other != args.strategy:
                try:
                    records[f"q3_sf10_{other}"], _ = q3.record(args, clock, torch, dist, False, strategy=other)
                except Exception as e:   # (an alternative strategy's record never takes the headline line down)
                    records[f"q3_sf10_{other}_error"] = f"{type(e).__name__}: {e}"
    if USE_DIST and os.environ.get("QHIP_BENCH_NO_SF100") != "1":
        # BASELINE configs[4]: Q3 with Zipf(1.1) join keys, every rank holding what ONE OF EIGHT ranks holds at SF100 (12.5
        # scale-factor units per rank: exactly SF100 at N = 8, SF25 / SF50 at N = 2 / 4) — the size at which a rank's local
        # work dwarfs the exchange; both strategies, the aggregate's LDS-table occupancy and the exchange's GB/s against xGMI
        try:
            za = argparse.Namespace(**vars(args))
            za.sf, za.skew, za.steps, za.warmup = float(os.environ.get("QHIP_BENCH_ZIPF_SF", 12.5 * world)), 1.1, max(2, min(args.steps, 5)), 1
            del q3
            zq = Q3(za, ctx, rank, world, args.strategy)
            zrec, _ = zq.record(za, clock, torch, dist, False)
            zrec["aggregate_table"] = zq.table_stats()
            other = "repartition" if args.strategy == "broadcast" else "broadcast"
            orec, _ = zq.record(za, clock, torch, dist, False, strategy=other)
            zrec["other_strategy"] = {k: orec[k] for k in ("strategy", "value", "ms_per_step", "exchange", "groups") if k in orec}
            records["q3_sf100_zipf"] = zrec
        except Exception as e:   # the headline line must not depend on this record
            records["q3_sf100_zipf_error"] = f"{type(e).__name__}: {e}"
    if not USE_DIST:
        # the GENERAL join layout — LDS-staged open-addressing regions + blocked hash filter, what north_star names — stays the
        # fallback of the dense (direct-address) layout TPC-H's integer keys take: its own record, so that it keeps a number
        os.environ["QHIP_JOIN_DENSE"] = "0"
        try:
            ctx.forget_plans()
            q3.plans = {}
            hrec, _ = q3.record(args, clock, torch, dist, False, extras=False)
            records["q3_sf10_hashed"] = {k: hrec[k] for k in ("workload", "value", "unit", "ms_per_step", "kernels", "device_ms_per_query", "host_waits_per_query", "groups")}
            records["q3_sf10_hashed"]["workload"] += "; QHIP_JOIN_DENSE=0: hashed join layouts only"
        except Exception as e:
            records["q3_sf10_hashed_error"] = f"{type(e).__name__}: {e}"
        finally:
            os.environ.pop("QHIP_JOIN_DENSE", None)
            ctx.forget_plans()
            q3.plans = {}
    cpu = None
    if with_cpu:
        c1, c3 = records["q1_sf10"]["cpu_baseline"], records["q3_sf10"]["cpu_baseline"]
        # the step on the CPU: SF10's rows of each query at the rate measured on its sample
        secs = SF10_LINEITEM_ROWS / c1["value"] + tot[2] / c3["value"]
        cpu = {"value": rows_step / secs, "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": "the step's two queries at the rates of their bounded samples (records.q1_sf10.cpu_baseline, records.q3_sf10.cpu_baseline: "
                         f"oracle/qoracle.c, 1 of {os.cpu_count()} host cores, 1 warm-up + median of 5 each)",
               "parts": {"q1_sf10_rows_per_s": c1["value"], "q3_sf10_lineitem_rows_per_s": c3["value"]}}
    line.update(value=rows_step * args.steps / elapsed, ms_per_step=elapsed / args.steps * 1e3, scaling="strong",
                config={"workload": "configs[2] TPC-H Q1 (aggregate list, q1.slt:5-12) over SF10's 59986052 lineitem rows + configs[3] TPC-H Q3 "
                                    "(q3.slt:1-24) at SF10, one pass of each per step; value = lineitem rows scanned by both / wall time; "
                                    "synthetic Arrow tables (SURVEY §8d recipes), HBM-resident",
                        "rows_per_step": {"q1_lineitem": SF10_LINEITEM_ROWS, "q3_customer": tot[0], "q3_orders": tot[1], "q3_lineitem": tot[2]},
                        "batch_rows": args.batch_rows, "q1_groups": records["q1_sf10"]["groups"], "q3_groups": tot[3],
                        "resident_bytes_per_gpu": q1.resident,
                        "parallelism": "one GPU" if not USE_DIST else f"every table sliced over {world} ranks; Q1 partial groups merged by all-gather, "
                                                                         f"Q3 joins: {args.strategy}"},
                roofline=records["q1_sf10"]["roofline"], cpu_baseline=cpu, records=records,
                execute_with_export_ms=(None if USE_DIST else records["q1_sf10"].get("execute_with_export_ms", 0) + records["q3_sf10"].get("execute_with_export_ms", 0)),
                cold_first_step_ms=(None if USE_DIST else records["q1_sf10"]["cold_table"]["cold_first_query_ms"] + records["q3_sf10"]["cold_table"]["cold_first_query_ms"]),
                setup_s={"q1_generate": q1.t_gen, "q1_upload_h2d": q1.t_upload, "h2d_GBps": q1.resident / max(q1.t_upload, 1e-9) / 1e9,
                         "h2d_GBps_warm_allocator": q1.resident / max(q1.t_upload_warm, 1e-9) / 1e9,
                         "note": "h2d_GBps = the process's FIRST upload (includes the allocator's first hipMalloc of the table's bytes); "
                                 "_warm_allocator = the same upload repeated into pooled memory = the transfer itself"})
    if "exchange" in records["q3_sf10"]:
        line["exchange"] = records["q3_sf10"]["exchange"]
    if world == 1 and not USE_DIST and not args.no_extra:
        try:
            records["partition"] = PartitionBench(ctx, q3.tabs).record(args, clock)
            records["filter_lineitem"] = FilterBench(args, ctx, q1.table).record(args, clock)
            mini = Q1(args, ctx, "q1_mini", 100_000_000, rank, world)
            records["q1_mini"], _ = mini.record(args, clock, with_cpu)
            if rank == 0:
                records["q1_mini"]["roofline"]["stream_read_GBps"] = ctx.measure_stream_read(4 << 30, 5)
            records["cpp_host"] = cpp_host_step(args)
        except Exception as e:   # the headline line must not depend on the extras
            records["extras_error"] = f"{type(e).__name__}: {e}"
    finish()


if __name__ == "__main__":
    main()
