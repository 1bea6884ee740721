"""Differential fuzzing of the HIP operators against the CPU oracle: random tables (NULLs, 1-byte and longer strings,
decimals, floats with NaN / -0, dates, ragged batches) through random plans (fused / separate filters, all join types with
one or two keys and residual filters, aggregates, sort, limit, projection with CASE / LIKE). Needs an MI355X.

    python tools/fuzz_plans.py [n_iterations] [first_seed]        (QHIP_FUZZ_REPEAT=3: executions per plan)
Prints every mismatch with its seed; exit code 1 if there was any."""
import decimal
import math
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa

import qurious_amd as q
from oracle import qoracle
from qurious_amd import JoinSide, JoinType, Operator
from qurious_amd import ScalarValue as S

D = decimal.Decimal
REPEAT = int(os.environ.get("QHIP_FUZZ_REPEAT", "3"))
DEC = pa.decimal128(15, 2)


def col(schema, name):
    return q.Column(name, schema.get_field_index(name))


def make_table(rng, prefix, n, nkeys):
    null_p = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    m = lambda: rng.random(n) < null_p   # noqa: E731
    flag_alphabet = ["A", "N", "R"] if rng.random() < 0.6 else ["AA", "N", "", "RRR"]
    flag_nulls = m() if rng.random() < 0.5 else np.zeros(n, dtype=bool)
    f = rng.normal(size=n)
    if n > 4:
        f[rng.integers(0, n, 2)] = np.nan
        f[rng.integers(0, n, 2)] = -0.0
    cols = {
        "k": pa.array(rng.integers(0, nkeys, n), type=pa.int64(), mask=m()),
        "flag": pa.array([flag_alphabet[v] for v in rng.integers(0, len(flag_alphabet), n)], type=pa.string(), mask=flag_nulls),
        "s": pa.array([("str%d" % v) * (1 + v % 3) for v in rng.integers(0, max(2, nkeys // 2), n)], type=pa.string(), mask=m()),
        "i": pa.array(rng.integers(-1000, 1000, n), type=pa.int32(), mask=m()),
        "d": pa.array([D(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)], type=DEC, mask=m()),
        "f": pa.array(f, type=pa.float64(), mask=m()),
        "day": pa.array(rng.integers(9000, 9050, n), type=pa.int32(), mask=m()).cast(pa.date32()),
        "b": pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=m()),
    }
    schema = pa.schema([pa.field(prefix + name, arr.type, True) for name, arr in cols.items()])
    batch = pa.RecordBatch.from_arrays(list(cols.values()), schema=schema)
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 4)))]))
    if rng.random() < 0.3 and len(cuts) > 2:
        cuts.insert(1, cuts[1])   # an empty batch in the middle
    batches = [batch.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])] or [batch]
    return schema, batches


def random_predicate(rng, schema, p):
    c = lambda name: col(schema, p + name)   # noqa: E731
    choices = [
        lambda: q.BinaryExpr(c("i"), Operator(int(rng.choice([int(Operator.Lt), int(Operator.GtEq), int(Operator.NotEq)]))), q.Literal(S.Int32(int(rng.integers(-500, 500))))),
        lambda: q.BinaryExpr(c("day"), Operator.Lt, q.CastExpr(q.Literal(S.Utf8("1994-09-%02d" % int(rng.integers(1, 29)))), pa.date32())),
        lambda: q.BinaryExpr(c("flag"), Operator.Eq, q.Literal(S.Utf8(str(rng.choice(["A", "N", "RRR", ""]))))),
        lambda: q.BinaryExpr(c("d"), Operator.Gt, q.CastExpr(q.Literal(S.Int64(int(rng.integers(-10**6, 10**6)))), DEC)),
        lambda: q.IsNotNull(c("k")),
        lambda: q.Like(bool(rng.random() < 0.3), c("s"), q.Literal(S.Utf8(str(rng.choice(["str1%", "%3str3%", "str_", "%"]))))),
        lambda: c("b"),
    ]
    e = choices[int(rng.integers(0, len(choices)))]()
    if rng.random() < 0.4:
        e = q.BinaryExpr(e, Operator(int(rng.choice([int(Operator.And), int(Operator.Or)]))), choices[int(rng.integers(0, len(choices)))]())
    return e


def string_exprs(rng, c):
    """round 4: expressions of Utf8 type (CASE over literals / columns / NULL, nested) and LIKE with a pattern column"""
    lit = lambda v: q.Literal(S.Utf8(v))   # noqa: E731
    pool = [
        q.CaseExpr([(c("b"), c("s"))], lit("none")),
        q.CaseExpr([(q.IsNull(c("flag")), lit(None))], c("flag")),
        q.CaseExpr([(q.BinaryExpr(c("i"), Operator.Gt, q.Literal(S.Int32(0))), lit("positive")), (q.IsNull(c("i")), c("s"))],
                   q.CaseExpr([(c("b"), lit(""))], c("flag"))),
        q.Like(bool(rng.random() < 0.5), c("s"), c("flag")),
        q.CaseExpr([(q.Like(False, c("s"), lit("%str1%")), lit("has a one, それ"))], c("s")),
    ]
    return [pool[int(k)] for k in rng.choice(len(pool), size=int(rng.integers(0, 4)), replace=False)]


def scan_of(rng, schema, batches, p, lazy=False):
    table = q.MemoryTable(schema, batches, lazy_upload=lazy)
    return q.Scan(schema, table, None, random_predicate(rng, schema, p) if rng.random() < 0.6 else None)


def random_aggregate(rng, input_plan, schema, p):
    c = lambda name: col(schema, p + name)   # noqa: E731
    keysets = [["flag"], ["k"], ["day"], ["flag", "day"], ["s"], ["k", "flag"], []]
    keys = keysets[int(rng.integers(0, len(keysets)))]
    revenue = q.BinaryExpr(c("d"), Operator.Mul, q.BinaryExpr(q.CastExpr(q.Literal(S.Int64(1)), pa.decimal128(20, 0)), Operator.Sub, c("d")))
    pool = [q.SumAggregateExpr(c("d"), DEC), q.CountAggregateExpr(c("i")), q.CountAggregateExpr(q.Literal(S.Int64(1))), q.MinAggregateExpr(c("i"), pa.int32()),
            q.MaxAggregateExpr(c("d"), DEC), q.SumAggregateExpr(c("f"), pa.float64()), q.AvgAggregateExpr(c("d"), DEC, pa.decimal128(19, 6)),
            q.SumAggregateExpr(revenue, pa.decimal128(38, 4)), q.MinAggregateExpr(c("day"), pa.date32()), q.MaxAggregateExpr(c("f"), pa.float64()),
            q.SumAggregateExpr(q.CaseExpr([(q.BinaryExpr(c("i"), Operator.Gt, q.Literal(S.Int32(0))), c("d"))], q.CastExpr(q.Literal(S.Int64(0)), DEC)), DEC)]
    aggs = [pool[int(k)] for k in rng.choice(len(pool), size=int(rng.integers(1, 5)), replace=False)]
    if not keys:
        # documented divergence: ungrouped MIN / MAX of floats when a batch holds a NaN depends on the batch
        # boundaries in the reference (that batch's contribution is discarded); not fuzzed
        aggs = [a for a in aggs if not (isinstance(a, (q.MinAggregateExpr, q.MaxAggregateExpr)) and pa.types.is_floating(a._return_type()))] or [pool[1]]
        return q.NoGroupingAggregate(None, input_plan, aggs), True
    return q.HashAggregate(None, input_plan, [c(k) for k in keys], aggs), True


def random_plan(rng):
    big = rng.random() < 0.2          # now and then: enough rows and groups for the device-side output assembly (>= 4096 groups)
    nl, nr = (int(rng.integers(50_000, 200_000)), int(rng.integers(20_000, 100_000))) if big else (int(rng.integers(0, 3000)), int(rng.integers(0, 3000)))
    nkeys = int(rng.choice([20_000, 60_000])) if big else int(rng.choice([3, 40, 2000]))
    ls, lb = make_table(rng, "l_", nl, nkeys)
    rs, rb = make_table(rng, "r_", nr, nkeys)
    kind = int(rng.integers(0, 7))
    unordered = False
    if kind == 0:   # aggregate over a (possibly filtered, possibly lazily uploaded) scan
        plan, unordered = random_aggregate(rng, scan_of(rng, ls, lb, "l_", lazy=rng.random() < 0.3), ls, "l_")
    elif kind == 1:  # separate Filter node, then maybe a projection
        plan = q.Filter(scan_of(rng, ls, lb, "l_"), random_predicate(rng, ls, "l_"))
        if rng.random() < 0.5:
            c = lambda name: col(ls, "l_" + name)   # noqa: E731
            plan = q.Projection(None, plan, [c("flag"), q.BinaryExpr(c("d"), Operator.Div, c("d")), q.Negative(c("i")), q.IsNull(c("s")),
                                              q.CaseExpr([(c("b"), c("f"))], q.Literal(S.Float64(0.5)))] + string_exprs(rng, c))
    elif kind == 5:  # ORDER BY (several keys of every type, top-N) straight over a scan, then maybe a window
        names = [str(x) for x in rng.choice(["d", "f", "s", "flag", "day", "i", "k", "b"], size=int(rng.integers(1, 4)), replace=False)]
        keys = [q.PhysicalSortExpr(col(ls, "l_" + name), q.SortOptions(bool(rng.random() < 0.5), bool(rng.random() < 0.5))) for name in names]
        plan = q.Sort(keys, scan_of(rng, ls, lb, "l_", lazy=rng.random() < 0.3), int(rng.integers(0, 200)) if rng.random() < 0.5 else None)
        if rng.random() < 0.5:
            plan = q.Limit(plan, int(rng.integers(0, 100)), int(rng.integers(0, 30)))
    elif kind == 6:  # cross join of two small inputs, aggregated or as is
        a = scan_of(rng, ls, [b.slice(0, min(b.num_rows, 40)) for b in lb[:2]], "l_")
        b2 = scan_of(rng, rs, [b.slice(0, min(b.num_rows, 30)) for b in rb[:3]], "r_")
        plan = q.CrossJoin(a, b2)
        if rng.random() < 0.5:
            plan, unordered = random_aggregate(rng, plan, plan.schema(), "l_")
    else:            # join, then something on top
        jt = JoinType(int(rng.integers(0, 6)))
        on_sets = [[("k", "k")], [("flag", "flag")], [("k", "k"), ("flag", "flag")], [("s", "s")], [("day", "day"), ("k", "k")]]
        on = [(col(ls, "l_" + a), col(rs, "r_" + b)) for a, b in on_sets[int(rng.integers(0, len(on_sets)))]]
        jf = None
        if rng.random() < 0.4:
            fschema = pa.schema([pa.field("l_i", pa.int32()), pa.field("r_i", pa.int32())])
            jf = q.JoinFilter(q.BinaryExpr(q.Column("l_i", 0), Operator.LtEq, q.Column("r_i", 1)), [(ls.get_field_index("l_i"), JoinSide.Left), (rs.get_field_index("r_i"), JoinSide.Right)], fschema)
        left, right = scan_of(rng, ls, lb, "l_"), scan_of(rng, rs, rb, "r_")
        if big and len(on) == 1 and "flag" in str(on[0][0]):
            on = [(col(ls, "l_k"), col(rs, "r_k"))]   # a 3-value key over 10^5 x 10^5 rows would produce billions of pairs (Semi / Anti joins
                                                      # emit none of them, but the oracle still walks every chain: minutes per plan)
        if kind == 4 and rng.random() < 0.5 and nl * nr < 2_000_000:
            plan = q.NestedLoopJoinExec.try_new(left, right, jt, jf)
        else:
            plan = q.HashJoinExec.try_new(left, right, jt, on, jf)
        js = plan.schema()
        top = int(rng.integers(0, 5))
        if top == 0:
            plan, unordered = random_aggregate(rng, plan, js, "l_")
        elif top == 1:
            keys = [q.PhysicalSortExpr(col(js, "l_" + name), q.SortOptions(bool(rng.random() < 0.5), bool(rng.random() < 0.5))) for name in rng.choice(["d", "f", "s", "flag", "day", "i", "k"], size=2, replace=False)]
            plan = q.Sort(keys, plan, int(rng.integers(0, 50)) if rng.random() < 0.5 else None)
        elif top == 2:
            plan = q.Limit(plan, int(rng.integers(0, 100)) if rng.random() < 0.7 else None, int(rng.integers(0, 50)))
        elif top == 4 and jt in (JoinType.Inner, JoinType.Left, JoinType.Right, JoinType.Full):
            # a projection over the join's (index-vector) output, strings computed from both sides
            c = lambda name: col(js, "l_" + name)   # noqa: E731
            plan = q.Projection(None, plan, [c("k"), col(js, "r_flag")] + string_exprs(rng, c))
    return plan, unordered


def norm(v):
    if isinstance(v, float):
        return ("nan",) if v != v else (round(v, 6), math.copysign(1.0, v) if v == 0 else 0)
    return v


def rows(batches, unordered):
    out = []
    for b in batches:
        # dates as day numbers: an all-NULL group's MIN is the seed i32::MAX (aggregate/mod.rs:60-84), which is not a valid date
        cols = [(c.cast(pa.int32()) if pa.types.is_date32(c.type) else c).to_pylist() for c in b.columns]
        out.extend(zip(*cols)) if b.num_columns else None
    out = [tuple(norm(v) for v in r) for r in out]
    return sorted(out, key=repr) if unordered else out


def close(a, b):
    if len(a) != len(b):
        return False
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            if x == y:
                continue
            if isinstance(x, tuple) and isinstance(y, tuple) and len(x) == 2 and len(y) == 2 and isinstance(x[0], float) and isinstance(y[0], float) and abs(x[0] - y[0]) <= 1e-6 * max(1.0, abs(y[0])):
                continue
            return False
    return True


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    q.get_context()
    bad = 0
    skipped = {}
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        try:
            plan, unordered = random_plan(rng)
        except Exception as e:   # plan construction itself
            print(f"seed {seed}: plan construction failed: {type(e).__name__}: {e}", flush=True)
            bad += 1
            continue
        got = want = None
        gerr = werr = None
        try:
            want_b = qoracle.execute(plan)
            want = rows(want_b, unordered)
        except Exception as e:
            werr = f"{type(e).__name__}: {e}"
        try:
            # several executions of the same plan: from the second on, hash joins under an aggregate / a build side leave their
            # output size on the device (remembered sizes, qhip_ctx_allow_deferred_sizes) and plans reuse what they learnt
            for rep in range(REPEAT):
                got_b = plan.execute()
                got_r = rows(got_b, unordered)
                if got is not None and not close(got_r, got):
                    print(f"seed {seed}: execution {rep + 1} differs from execution 1 plan={type(plan).__name__}", flush=True)
                    bad += 1
                got = got_r if got is None else got
        except Exception as e:
            gerr = f"{type(e).__name__}: {e}"
        if (gerr is None) != (werr is None):
            if gerr and ("not accelerated" in gerr or "not supported" in gerr or "Unsupported" in gerr):
                skipped[gerr[:90]] = skipped.get(gerr[:90], 0) + 1
                continue   # a documented QHIP_UNSUPPORTED (the shim would fall back to the CPU node); counted and listed below
            print(f"seed {seed}: error mismatch: hip={gerr} oracle={werr} plan={type(plan).__name__}", flush=True)
            bad += 1
        elif gerr is None:
            same_batches = unordered or [b.num_rows for b in got_b] == [b.num_rows for b in want_b]
            if not close(got, want) or not same_batches:
                print(f"seed {seed}: RESULT MISMATCH plan={type(plan).__name__} rows hip={len(got)} oracle={len(want)} batches hip={[b.num_rows for b in got_b][:8]} oracle={[b.num_rows for b in want_b][:8]}", flush=True)
                for k, (x, y) in enumerate(zip(got, want)):
                    if x != y:
                        print("   first difference at row", k, x, y, flush=True)
                        break
                bad += 1
        if seed % 25 == 0:
            print(f"... seed {seed} done, {bad} problems so far", flush=True)
    for reason, count in sorted(skipped.items()):
        print(f"   {count} plans not accelerated: {reason}")
    print(f"{n} plans, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except Exception:
        traceback.print_exc()
        sys.exit(2)
